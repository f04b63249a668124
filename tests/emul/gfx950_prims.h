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
template<class T> inline T stream_load(const T* q) { return *q; }
template<class T> inline void stream_store(T* q, T v) { *q = v; }
#define MHH_RAW_DMA 0
inline unsigned lds_address(void*) { return 0; }
inline unsigned uniform_u32(unsigned v) { return v; }
template<int PB> inline void lds_dma_sv(const void*, unsigned, unsigned) {}
template<class T> inline T uniform_load(const T* table, int idx) { return table[idx]; }
}
