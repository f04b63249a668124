// wave_reduce.h -- block-wide max of non-negative values and its hand-off to one device word (gfx950).
// 64-wide wavefront shuffles, then LDS across the block's waves, then ONE integer atomicMax per block:
// a non-negative IEEE value orders like its bit pattern taken as an unsigned integer.
#pragma once
#include <hip/hip_runtime.h>

namespace mhh
{
template<class TF> struct Bits;
template<> struct Bits<double> { using U = unsigned long long; static __device__ __forceinline__ U to(double v) { return (U)__double_as_longlong(v); } };
template<> struct Bits<float>  { using U = unsigned int;       static __device__ __forceinline__ U to(float v)  { return __float_as_uint(v); } };

// Every thread of the (NW waves x 64 lanes) block must call this; thread (0,0) publishes the block maximum.
template<class TF, int NW>
__device__ __forceinline__ void block_max_publish(TF m, typename Bits<TF>::U* __restrict__ out)
{
    for (int off = 32; off > 0; off >>= 1)
    {
        const TF o = __shfl_down(m, off, 64);
        m = (m < o) ? o : m;
    }
    __shared__ TF part[NW];
    const int wave = threadIdx.y;              // blockDim.x == 64: one wavefront per y row of the block
    if (threadIdx.x == 0) part[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0)
    {
        TF b = part[0];
        for (int n=1; n<NW; ++n) b = (b < part[n]) ? part[n] : b;
        atomicMax(out, Bits<TF>::to(b));
    }
}
}
