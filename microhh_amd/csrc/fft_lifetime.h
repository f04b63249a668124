// fft_lifetime.h -- one rocFFT library set-up for the whole process, shared by every plan type (k_pres.hip: single-rank plans,
// k_slab.hip: slab plans). rocfft_setup() runs once, before the first plan is made; rocfft_cleanup() is never called while the
// process lives: a count kept per translation unit let the last single-rank plan tear rocFFT down underneath a slab plan
// that was still alive.
#pragma once
#include <mutex>
#include <rocfft/rocfft.h>

namespace mhh
{
inline void fft_acquire()
{
    static std::once_flag once;
    std::call_once(once, [] { rocfft_setup(); });
}
}
