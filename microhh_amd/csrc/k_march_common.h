// k_march_common.h -- tiling shared by the k-marching kernels (k_march.hip, k_visc.hip): 64 x NJ column tiles that walk
// up in k over chunks of kc levels, dealt to the 8 XCDs so that tiles sharing halos run next to each other on one L2.
#pragma once
#include "k_common.h"

namespace mhh
{
struct MarchTiling { int nbx, nby, nkc, sr, ns, kc; };

__device__ __forceinline__ bool decode_march(const MarchTiling& t, unsigned L, int& bx, int& by, int& kc)
{
    // XCD-aware order as decode_tile (k_common.h): a unit = one strip of tiles over one k-chunk; units are dealt to the
    // XCDs round-robin, so the tiles that share halos run next to each other on one L2.
    const int xcd = L & 7u;
    const unsigned tt = L >> 3;
    const unsigned per_unit = (unsigned)t.sr * t.nbx;
    const unsigned round = tt / per_unit;
    unsigned r = tt - round * per_unit;
    const int unit = (int)round * 8 + xcd;
    if (unit >= t.ns * t.nkc) return false;
    const int strip = unit % t.ns;
    kc = unit / t.ns;
    const int byl = (int)(r / t.nbx);
    bx = (int)(r - (unsigned)byl * t.nbx);
    by = strip * t.sr + byl;
    return by < t.nby;
}

// strips of sr tile rows; (strip, k-chunk) units are dealt round-robin to the XCDs; sr shrinks on thin slabs so that all
// 8 XCDs get work
inline MarchTiling make_march_tiling(const mhh_grid* g, int NJ, int kc)
{
    MarchTiling t;
    t.nbx = (g->imax + 63)/64; t.nby = (g->jmax + NJ-1)/NJ;
    t.kc = kc; t.nkc = (g->kmax + t.kc - 1)/t.kc;
    t.sr = (MHH_STRIP_ROWS + NJ-1)/NJ;
    if (t.sr * 8 > t.nby * t.nkc) t.sr = (t.nby * t.nkc) / 8;
    if (t.sr < 1) t.sr = 1;
    t.ns = (t.nby + t.sr-1)/t.sr;
    return t;
}
inline unsigned march_blocks(const MarchTiling& t)
{
    const int units = t.ns * t.nkc;
    return 8u * (unsigned)((units + 7)/8) * (unsigned)t.sr * t.nbx;
}
}
