// gfx950_prims.h -- the few CDNA4-specific primitives the kernels use directly.
#pragma once
#include <hip/hip_runtime.h>

namespace mhh
{
// Asynchronous global -> LDS copy of 16 bytes per lane (global_load_lds_dwordx4): every active lane supplies its own
// global address; the data lands at `lds_wave_base + lane*16` (a wave-uniform base, contiguous by lane -- not a
// per-lane scatter). No VGPR is written; completion is tracked by vmcnt.
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The 4-byte form (global_load_lds_dword): data lands at `lds_wave_base + lane*4`; needs 4-byte alignment only.
__device__ __forceinline__ void lds_dma4(const void* gsrc, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
// Wait until all of this wave's vector-memory operations (loads, stores, LDS-DMA) have completed. Inline asm on
// purpose: the compiler may not elide or move it (MI355X_MICROARCH.md, "Compiler hazard").
__device__ __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
}
