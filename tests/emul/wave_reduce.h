// tests/emul/wave_reduce.h -- TEST-ONLY override of microhh_amd/csrc/wave_reduce.h for the CPU emulation build.
// The emulator runs a block's threads sequentially with thread (0,0) LAST, so a running maximum per block is
// complete when that thread publishes it.
#pragma once
#include <hip/hip_runtime.h>
namespace mhh
{
template<class TF> struct Bits;
template<> struct Bits<double> { using U = unsigned long long; };
template<> struct Bits<float>  { using U = unsigned int; };
template<class TF, int NW>
inline void block_max_publish(TF m, typename Bits<TF>::U* out)
{
    static thread_local TF acc = 0;
    const bool first = (threadIdx.x == blockDim.x-1 && threadIdx.y == blockDim.y-1 && threadIdx.z == blockDim.z-1);
    if (first) acc = m; else acc = (acc < m) ? m : acc;
    if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0)
    {
        TF cur; memcpy(&cur, out, sizeof(TF));
        if (cur < acc) memcpy(out, &acc, sizeof(TF));
    }
}
}
