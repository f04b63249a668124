// gfx950_prims.h -- the few CDNA4-specific primitives the kernels use directly.
#pragma once
#include <hip/hip_runtime.h>

namespace mhh
{
// Asynchronous global -> LDS copy of 16 bytes per lane (global_load_lds_dwordx4): every active lane supplies its own
// global address; the data lands at `lds_wave_base + lane*16` (a wave-uniform base, contiguous by lane -- not a
// per-lane scatter). No VGPR is written; completion is tracked by vmcnt.
//
// Issued as inline asm, not through __builtin_amdgcn_global_load_lds: the compiler models the builtin as a store to LDS
// that any later ds_read may alias (the ring slot is a run-time index), so it puts an s_waitcnt vmcnt(0) in front of the
// first LDS read after the copy was issued -- the copy of the NEXT plane then never overlaps the compute on the current
// ones, which is the whole point of the ring. Here the ordering is the kernels' own: a copy is issued after the barrier
// that retires the slot's last readers, and wait_vmem() + barrier stand between the copy and the slot's first reader.
// RAW = false is the builtin form (2i5+smag2 -1.5 %, advec_4+diff_4 -10 % at 512x256x256 with RAW). With the 64-bit vector
// address of this first raw form the fp32 kernels, at four waves per SIMD, measured 11 % SLOWER (gabls1 1024x1024x256);
// the scalar-base form below (lds_dma_sv) is the one all kernels use now, fp32 included (-1 % there).
// (-DMHH_DMA_BUILTIN forces the builtin everywhere for A/B runs.)
__device__ __forceinline__ unsigned lds_address(void* lds_wave_base)
{
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base);
}
template<bool RAW> __device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_wave_base)
{
#ifdef MHH_DMA_BUILTIN
    constexpr bool raw = false;
#else
    constexpr bool raw = RAW;
#endif
    if constexpr (!raw)
    {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                         (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
    }
    else
    {
        unsigned m0_saved;     // m0 is a reserved register: put back what the compiler may have parked there
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_saved) : "v"(gsrc), "s"(lds_address(lds_wave_base)) : "memory");
    }
}
// The 4-byte form (global_load_lds_dword): data lands at `lds_wave_base + lane*4`; needs 4-byte alignment only.
template<bool RAW> __device__ __forceinline__ void lds_dma4(const void* gsrc, void* lds_wave_base)
{
#ifdef MHH_DMA_BUILTIN
    constexpr bool raw = false;
#else
    constexpr bool raw = RAW;
#endif
    if constexpr (!raw)
    {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                         (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
    }
    else
    {
        unsigned m0_saved;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_saved) : "v"(gsrc), "s"(lds_address(lds_wave_base)) : "memory");
    }
}
// The raw form with the address split the way the hardware takes it: a wave-uniform 64-bit base in scalar registers plus a
// per-lane 32-bit byte offset (which for a tile copy is the same on every level: computed once), and the wave's LDS
// address as a scalar -- no vector ALU work per piece at all.
#define MHH_RAW_DMA 1
__device__ __forceinline__ unsigned uniform_u32(unsigned wave_uniform_value) { return __builtin_amdgcn_readfirstlane(wave_uniform_value); }
template<int PB> __device__ __forceinline__ void lds_dma_sv(const void* uniform_base, unsigned lane_byte_offset, unsigned lds_wave_address)
{
    unsigned m0_saved;
    if constexpr (PB == 16)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_saved) : "v"(lane_byte_offset), "s"(uniform_base), "s"(lds_wave_address) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_saved) : "v"(lane_byte_offset), "s"(uniform_base), "s"(lds_wave_address) : "memory");
}
// The same with the LDS address as the sum of two scalars, added inside the statement: `lds_wave_base` (this wave's lane-0
// address in the kernel's LDS array) + `lds_offset` (the destination tile, usually a compile-time constant, which then costs
// one s_mov and no register across the loop). Written as C the sum is loop-invariant, gets hoisted for every (tile, piece)
// pair and the ~20 sums are spilled to vector-register lanes and read back with v_readlane every level.
template<int PB> __device__ __forceinline__ void lds_dma_sv2(const void* uniform_base, unsigned lane_byte_offset, unsigned lds_wave_base, unsigned lds_offset)
{
    unsigned m0_saved;
    // wave-uniform by contract; what the compiler nevertheless holds in vector registers is moved (a no-op otherwise)
    lds_offset = __builtin_amdgcn_readfirstlane(lds_offset);
    const unsigned long long ub = (unsigned long long)uniform_base;
    // (the builtin returns int: without the casts a low word with bit 31 set sign-extends over the high word)
    uniform_base = (const void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ub >> 32)) << 32) |
                                 (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ub));
#ifdef MHH_DMA_M0_SAVE
    if constexpr (PB == 16)
        asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %4\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_saved) : "v"(lane_byte_offset), "s"(uniform_base), "s"(lds_wave_base), "s"(lds_offset) : "memory", "scc");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %4\n\ts_nop 2\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_saved) : "v"(lane_byte_offset), "s"(uniform_base), "s"(lds_wave_base), "s"(lds_offset) : "memory", "scc");
#else
    // M0 is left holding the LDS address and DECLARED clobbered: the compiler saves and restores it around the statement only
    // where it has a use of its own for M0 (in these kernels it has none: on gfx9 the ds_ instructions do not read it; no movrel,
    // GWS, sendmsg or LDS-DMA builtin beside this primitive), so the two s_mov that an unconditional save / restore put around every
    // copy (18 of a level's ~180 scalar instructions; -DMHH_DMA_M0_SAVE brings them back) are gone without an assumption about
    // code generation
    (void)m0_saved;
    if constexpr (PB == 16)
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %0, %1"
                     : : "v"(lane_byte_offset), "s"(uniform_base), "s"(lds_wave_base), "s"(lds_offset) : "memory", "scc", "m0");
    else
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 2\n\tglobal_load_lds_dword %0, %1"
                     : : "v"(lane_byte_offset), "s"(uniform_base), "s"(lds_wave_base), "s"(lds_offset) : "memory", "scc", "m0");
#endif
}
// Pin a wave-uniform value in scalar registers of its own. Kernel arguments arrive by merged s_load_dwordx8/x16, and the
// register allocator keeps, spills and restores such a tuple as ONE value: to use one pointer of a spilled argument struct
// it restores all sixteen dwords, each with a v_readlane_b32 -- a vector-ALU instruction. A value passed through here is a
// separate live range (1-2 registers), and the tuple dies at the top of the kernel.
#ifndef MHH_NO_PIN
template<class T> __device__ __forceinline__ T sgpr(T x) { asm volatile("" : "+s"(x)); return x; }
#else
template<class T> __device__ __forceinline__ T sgpr(T x) { return x; }
#endif
// Load of a wave-uniform element of a read-only table (per-level metrics, base-state profiles) through the constant
// address space, i.e. as an s_load on the scalar cache, tracked by lgkmcnt. A plain load of such an element inside a loop
// that also stores is emitted as a VECTOR load (the compiler cannot prove the table is not clobbered), and its
// s_waitcnt vmcnt(0) then also waits for every LDS-DMA copy and prefetch in flight -- the latency the marching kernels
// are built to hide. Only for memory that no thread writes during the kernel (the scalar cache is not coherent with stores).
template<class T> __device__ __forceinline__ T uniform_load(const T* table, int idx)
{
    typedef const T __attribute__((address_space(4)))* const_ptr;
    return ((const_ptr)(table))[idx];
}
// Eight consecutive wave-uniform elements in one scalar load (s_load_dwordx16 / x8)
template<class T> struct alignas(16) Uniform8 { T v[8]; };
template<class T> __device__ __forceinline__ Uniform8<T> uniform_load8(const T* table)
{
    typedef T vec8 __attribute__((ext_vector_type(8)));
    typedef const vec8 __attribute__((address_space(4)))* const_ptr;
    const vec8 x = *(const_ptr)(table);
    Uniform8<T> r;
#pragma unroll
    for (int n=0; n<8; ++n) r.v[n] = x[n];
    return r;
}
// The kernel's FIRST argument as memory: the kernel-argument segment is constant memory behind the scalar cache, so a kernel can
// re-read a by-value argument with s_load where it needs it instead of holding it in registers from its first instruction on.
// (Taking the address of the parameter itself would make the compiler copy it to scratch.)
template<class T> __device__ __forceinline__ const T* first_kernarg(const T&)
{
    return (const T*)__builtin_amdgcn_kernarg_segment_ptr();
}
// Loads / stores of data touched once per kernel (the tendencies of the marching kernels): the non-temporal hint lets them
// stream past L2 instead of evicting the planes that neighbouring tiles re-read.
template<class T> __device__ __forceinline__ T stream_load(const T* q) { return __builtin_nontemporal_load(q); }
template<class T> __device__ __forceinline__ void stream_store(T* q, T v) { __builtin_nontemporal_store(v, q); }
// Global loads / stores through a wave-uniform base pointer plus a per-lane 32-bit byte offset: the saddr + voffset form of
// global_load / global_store, with no 64-bit vector address arithmetic. The explicit global address space matters: a pointer
// that went through sgpr() is opaque to the compiler's address-space inference and would be accessed with flat_* instructions.
#ifndef MHH_GLOAD_PLAIN
template<class T> __device__ __forceinline__ const T __attribute__((address_space(1)))* global_at(const T* uniform_base, unsigned lane_byte_offset)
{
    typedef const char __attribute__((address_space(1)))* gcp;
    typedef const T __attribute__((address_space(1)))* gtp;
    return (gtp)((gcp)uniform_base + lane_byte_offset);
}
#else
template<class T> __device__ __forceinline__ const T* global_at(const T* uniform_base, unsigned lane_byte_offset)
{
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(uniform_base) + lane_byte_offset);
}
#endif
template<class T> __device__ __forceinline__ T gload(const T* uniform_base, unsigned lane_byte_offset) { return *global_at(uniform_base, lane_byte_offset); }
template<class T> __device__ __forceinline__ T gload_stream(const T* uniform_base, unsigned lane_byte_offset) { return __builtin_nontemporal_load(global_at(uniform_base, lane_byte_offset)); }
template<class T> __device__ __forceinline__ void gstore(T* uniform_base, unsigned lane_byte_offset, T v)
{
#ifndef MHH_GLOAD_PLAIN
    typedef T __attribute__((address_space(1)))* gtp; *(gtp)global_at(uniform_base, lane_byte_offset) = v;
#else
    *const_cast<T*>(global_at(uniform_base, lane_byte_offset)) = v;
#endif
}
template<class T> __device__ __forceinline__ void gstore_stream(T* uniform_base, unsigned lane_byte_offset, T v)
{
#ifndef MHH_GLOAD_PLAIN
    typedef T __attribute__((address_space(1)))* gtp; __builtin_nontemporal_store(v, (gtp)global_at(uniform_base, lane_byte_offset));
#else
    __builtin_nontemporal_store(v, const_cast<T*>(global_at(uniform_base, lane_byte_offset)));
#endif
}
// Opaque re-definition of a per-lane value (no instruction): what is computed from it cannot be hoisted above this point.
__device__ __forceinline__ void keep_vgpr(unsigned& x) { asm volatile("" : "+v"(x)); }
// The same for a value of any register type: it lives in vector registers from here on (e.g. a scalar that several VOP3
// instructions need beside another scalar operand: one v_mov here instead of one per use)
template<class T> __device__ __forceinline__ void pin_vgpr(T& x) { asm volatile("" : "+v"(x)); }
// Orders the LDS traffic of ONE wave: what its lanes wrote before this point is what its lanes read after it. The LDS serves a
// wave's instructions in order, so no s_barrier is involved -- the fence only keeps the compiler from moving accesses across.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic, not for its global loads and stores.
// __syncthreads() is a release / acquire fence on ALL memory plus the barrier, i.e. s_waitcnt vmcnt(0) -- a kernel that requests
// the next batch from global memory and then synchronises its waves on LDS data would wait out the full memory latency at that
// barrier, and every store before one as well. (Global data written by one thread and read by ANOTHER needs __syncthreads.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// v_rcp_f64 / v_rcp_f32: the hardware's reciprocal seed (refined by the caller)
__device__ __forceinline__ double recip_seed(double x) { return __builtin_amdgcn_rcp(x); }
__device__ __forceinline__ float  recip_seed(float x)  { return __builtin_amdgcn_rcpf(x); }
// No instruction is scheduled across this point (bounds the live ranges the instruction scheduler creates by hoisting loads).
#ifndef MHH_NO_SCHED_FENCE
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
#else
__device__ __forceinline__ void sched_fence() {}
#endif
// Wait until all of this wave's vector-memory operations (loads, stores, LDS-DMA) have completed. Inline asm on
// purpose: the compiler may not elide or move it (MI355X_MICROARCH.md, "Compiler hazard").
// The builtin behind it (s_waitcnt vmcnt(0), other counters untouched) tells the compiler's own wait-count bookkeeping
// the same thing: without it, values loaded in one loop iteration and first used in the next (the prefetched tendencies)
// get a compiler-inserted vmcnt(0) at that use -- AFTER the next level's copies have been issued, draining them.
__device__ __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_waitcnt(0x0F70); }
// The value lane `lane` (wave-uniform) holds of a per-lane value, as a wave-uniform value: a wave computes up to 64 different uniform
// values with ONE vector instruction sequence (lane l works value l) and hands them out one per iteration with two v_readlane --
// e.g. a division by a per-level constant, which has no scalar form. `same` = the value computed the ordinary way: what the CPU
// emulation of the kernels returns (its lanes are independent threads); unused on the device.
__device__ __forceinline__ double value_of_lane(double v, int lane, double /*same*/)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float value_of_lane(float v, int lane, float /*same*/)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// Square root of a double known to lie in [2^-767, +inf): the compiler's own expansion of sqrt on this target (v_rsq_f64 seed, two
// Goldschmidt steps, two residual corrections -- the same ten instructions on the same values, hence the same bits) without the
// seven instructions around it that serve arguments outside that range: the 2^256 scaling of tiny arguments (compare, select,
// two ldexp) and the pass-through of 0 and +inf (class test, two selects). An argument of +inf gives NaN here, not +inf.
__device__ __forceinline__ double sqrt_in_range(double x)
{
    const double y  = __builtin_amdgcn_rsq(x);
    const double g0 = x*y, h0 = y*0.5;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-g1, g1, x);
    const double g2 = __builtin_fma(d0, h1, g1);
    const double d1 = __builtin_fma(-g2, g2, x);
    return __builtin_fma(d1, h1, g2);
}
__device__ __forceinline__ float sqrt_in_range(float x) { return __builtin_sqrtf(x); }
// true if the predicate holds on any lane of the wave (wave-uniform)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
}
