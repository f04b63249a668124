// Shared bits of the oracle/_ref harness: thin extern "C" wrappers that call the reference's own
// kernels (anonymous-namespace templates) after `#include`-ing the reference translation unit in place.
// No reference source is copied; nothing is stubbed: unused reference member functions (which reference
// Grid/Stats/Input symbols we cannot build) are discarded by -fvisibility=hidden + --gc-sections.
#pragma once
#include "../../include/mhh_hip.h"
#define REF_API extern "C" __attribute__((visibility("default")))
template<class TF> static inline const TF* CP(const void* p) { return static_cast<const TF*>(p); }
template<class TF> static inline TF* MP(void* p) { return static_cast<TF*>(p); }
#define GRID_BOUNDS(g) (g)->istart, (g)->iend, (g)->jstart, (g)->jend, (g)->kstart, (g)->kend
