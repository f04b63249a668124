// k_march_common.h -- tiling shared by the k-marching kernels (k_march.hip, k_visc.hip): 64 x NJ column tiles that walk
// up in k over chunks of kc levels, dealt to the 8 XCDs so that tiles sharing halos run next to each other on one L2.
#pragma once
#include <cstdint>
#include "k_common.h"
#include <gfx950_prims.h>

namespace mhh
{
// rows [jbase, jlim) are worked, in tiles from jbase; optionally a SECOND range [jbase2, jlim2) in the same launch (the two edge
// strips of a slab once its north-south halos have arrived: one launch instead of two): tile rows by >= nby1 belong to it
struct MarchTiling { int nbx, nby, nkc, sr, ns, kc, jbase, jlim, nby1, jbase2, jlim2; };
__device__ __forceinline__ void march_tile_rows(const MarchTiling& t, int by, int NJ, int& j0, int& jlim)
{
    if (by < t.nby1) { j0 = t.jbase + by*NJ; jlim = t.jlim; }
    else             { j0 = t.jbase2 + (by - t.nby1)*NJ; jlim = t.jlim2; }
}

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
// tw = cells of a row per tile (a wave's 64 lanes times the cells per lane)
inline MarchTiling make_march_tiling(const mhh_grid* g, int NJ, int kc, int j0 = -1, int j1 = -1, int tw = 64, int j2 = -1, int j3 = -1)
{
    MarchTiling t;
    t.jbase = (j0 < 0) ? g->jstart : j0; t.jlim = (j1 < 0) ? g->jend : j1;
    t.nbx = (g->imax + tw-1)/tw; t.nby1 = (t.jlim - t.jbase + NJ-1)/NJ; t.nby = t.nby1;
    t.jbase2 = t.jlim2 = t.jlim;
    if (j2 >= 0 && j3 > j2) { t.jbase2 = j2; t.jlim2 = j3; t.nby += (j3 - j2 + NJ-1)/NJ; }
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

// LDS-DMA copy of one TI x TJ tile of a plane (origin gi0, gj0 in grid cells) into an LDS slot, in pieces of PB bytes per
// lane: PB = 16 (global_load_lds_dwordx4; rows and origin 16-byte aligned) or PB = 4 (global_load_lds_dword; any layout,
// four times the instructions). Piece e = tid + n*NT covers words [tw, tw+PW) of tile row tj; lanes whose piece falls
// outside the tile or the array sit out. lds_dma16 / lds_dma4 come from <gfx950_prims.h> (included by the kernels).
template<class TF, int PB, int TI, int TJ, int NT>
struct TileCopy
{
    static constexpr int EW = (int)sizeof(TF) / 4, PW = PB / 4;
    static constexpr int PPR = TI*EW / PW, NP = PPR*TJ, NLD = (NP + NT - 1) / NT;
    static_assert(PB == 4 || PB == 16, "piece size");
    static_assert((TI*EW) % PW == 0, "tile row must be a whole number of pieces");
    // SV: scalar-base + lane-offset form of the raw copy (see gfx950_prims.h); -DMHH_DMA_NO_SV: the 64-bit-address form (fp64) / builtin (fp32)
#ifdef MHH_DMA_NO_SV
    static constexpr bool SV = false;
#else
    static constexpr bool SV = (MHH_RAW_DMA != 0);
#endif
    int off[NLD]; bool ok[NLD]; int wave_e0; unsigned wave_lds;
    __device__ __forceinline__ void init(int tid, int gi0, int gj0, int icells, int jcells)
    {
        wave_e0 = tid & ~63;
        wave_lds = uniform_u32((unsigned)(wave_e0*PW*4));       // byte offset of this wave's lanes within a slot, as a scalar
#pragma unroll
        for (int n=0; n<NLD; ++n)
        {
            const int e = tid + n*NT;
            const int tj = e / PPR, tw = (e - tj*PPR)*PW;
            const int gw = gi0*EW + tw, gj = gj0 + tj;
            ok[n] = (e < NP) && (gw >= 0) && (gw + PW <= icells*EW) && (gj >= 0) && (gj < jcells);
            off[n] = ok[n] ? gw + gj*icells*EW : 0;
        }
    }
    __device__ __forceinline__ void copy(const TF* __restrict__ plane, TF* __restrict__ lds) const
    {
        const uint32_t* __restrict__ pl = reinterpret_cast<const uint32_t*>(plane);
        uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(lds);
#pragma unroll
        for (int n=0; n<NLD; ++n)
            if (ok[n])
            {
                if constexpr (SV)       lds_dma_sv<PB>(plane, (unsigned)off[n]*4u, lds_address(lds) + (wave_lds + (unsigned)(n*NT*PW*4)));
                else if constexpr (PB == 16) lds_dma16<(sizeof(TF) == 8)>(pl + off[n], dst + (size_t)(wave_e0 + n*NT)*PW);
                else                    lds_dma4<(sizeof(TF) == 8)>(pl + off[n], dst + (size_t)(wave_e0 + n*NT)*PW);
            }
    }
};
}
