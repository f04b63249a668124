// tests/emul/gfx950_prims.h -- TEST-ONLY override of microhh_amd/csrc/gfx950_prims.h for the CPU emulation build.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
namespace mhh
{
template<bool RAW = false> inline void lds_dma16(const void* gsrc, void* lds_wave_base)
{
    const unsigned lane = (threadIdx.x + threadIdx.y*blockDim.x + threadIdx.z*blockDim.x*blockDim.y) & 63u;
    std::memcpy(static_cast<char*>(lds_wave_base) + lane*16, gsrc, 16);
}
template<bool RAW = false> inline void lds_dma4(const void* gsrc, void* lds_wave_base)
{
    const unsigned lane = (threadIdx.x + threadIdx.y*blockDim.x + threadIdx.z*blockDim.x*blockDim.y) & 63u;
    std::memcpy(static_cast<char*>(lds_wave_base) + lane*4, gsrc, 4);
}
inline void wait_vmem() {}

inline bool wave_any(bool) { return true; }
inline double value_of_lane(double, int, double same) { return same; }
inline float value_of_lane(float, int, float same) { return same; }
inline double sqrt_in_range(double x) { return __builtin_sqrt(x); }
inline float sqrt_in_range(float x) { return __builtin_sqrtf(x); }
template<class T> inline T stream_load(const T* q) { return *q; }
template<class T> inline void stream_store(T* q, T v) { *q = v; }
#define MHH_RAW_DMA 0
inline unsigned lds_address(void*) { return 0; }
inline unsigned uniform_u32(unsigned v) { return v; }
template<int PB> inline void lds_dma_sv(const void*, unsigned, unsigned) {}
template<int PB> inline void lds_dma_sv2(const void*, unsigned, unsigned, unsigned) {}
template<class T> inline T sgpr(T x) { return x; }
inline void sched_fence() {}
inline void lds_barrier() { __syncthreads(); }
inline void wave_sync() { __syncthreads(); }     // every thread of the block reaches the same wave_sync calls (uniform control flow)
inline double recip_seed(double x) { return 1./x; }
inline float  recip_seed(float x)  { return 1.f/x; }
inline void keep_vgpr(unsigned&) {}
template<class T> inline void pin_vgpr(T&) {}
template<class T> inline T gload(const T* b, unsigned o) { return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(b) + o); }
template<class T> inline T gload_stream(const T* b, unsigned o) { return gload(b, o); }
template<class T> inline void gstore(T* b, unsigned o, T v) { *reinterpret_cast<T*>(reinterpret_cast<char*>(b) + o) = v; }
template<class T> inline void gstore_stream(T* b, unsigned o, T v) { gstore(b, o, v); }
template<class T> inline T uniform_load(const T* table, int idx) { return table[idx]; }
template<class T> struct alignas(16) Uniform8 { T v[8]; };
template<class T> inline Uniform8<T> uniform_load8(const T* table) { Uniform8<T> r; for (int n=0; n<8; ++n) r.v[n] = table[n]; return r; }
template<class T> inline const T* first_kernarg(const T& first_argument) { return &first_argument; }

}
