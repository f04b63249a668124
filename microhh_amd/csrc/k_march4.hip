// k_march4.hip -- advec_4 + diff_4 for u, v, w in ONE pass as a k-marching LDS kernel (src/advec_4.cxx:88-486,
// src/diff_4.cxx:41-173), the 4th-order sibling of k_march.hip.
//
// The one-thread-per-cell form (Rhs44Op, k_rhs.hip) issues 175 vector loads per cell and is bound by the L1/TA pipe
// (3.3 ms at 512x256x256 fp64 for 72 B/cell of algorithmic traffic). Here a 64 x NJ block walks up in k with the planes
// that are read with horizontal offsets in LDS (halo 3): u, v at k-2..k+1 (the w equation interpolates them vertically
// at x / y offsets), w at k-1..k+2 (the u and v equations interpolate it horizontally at four levels); the 7-level
// column of each thread's own u, v, w sits in registers. The arithmetic is the view-generic code of cell_ops.h
// (advec4_mom_v, diff4_v), i.e. literally the expressions of the cell kernel: same bits.
//
// LDS budget: u and v keep exactly their four planes -- the w equation, the only reader of the oldest one, is computed
// first, then a barrier, then the copy of plane k+2 is issued into that slot and lands while the u and v equations are
// computed; w has a fifth slot for its copy. 13 planes of 70 x (NJ+6) doubles = 72.8 KB: two blocks per CU.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include "k_common.h"
#include "k_march_common.h"
#include <gfx950_prims.h>

using namespace mhh;

#ifdef MHH_FMA_BUILD     // the named FMA build (build.py): its kernels carry their own name in profiler output
#define rhs44_march_kernel rhs44_march_fma_kernel
#endif

namespace
{
template<class TF> struct March4Fields
{
    const TF* __restrict__ u; const TF* __restrict__ v; const TF* __restrict__ w;
    TF* __restrict__ ut; TF* __restrict__ vt; TF* __restrict__ wt;
    TF visc;
};

// field view of the marching kernel: horizontal offsets from the LDS plane of that level, the own column from registers
template<class TF, int TI> struct MarchView
{
    const TF* pl[5];                 // plane pointers (already at this thread's cell) for level offsets -2..+2
    const TF (&win)[7];              // own column, level offsets -3..+3
    template<int DI, int DJ, int DK> __device__ __forceinline__ TF at() const
    {
        if constexpr (DI == 0 && DJ == 0) return win[3+DK];
        else return pl[DK+2][DI + DJ*TI];
    }
};

template<class TF> __device__ __forceinline__ void shift7(TF (&w)[7], TF nw)
{
    w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = w[4]; w[4] = w[5]; w[5] = w[6]; w[6] = nw;
}

#ifndef MHH_MARCH4_OCC
#define MHH_MARCH4_OCC 2
#endif
#ifndef MHH_MARCH4_KC
#define MHH_MARCH4_KC 64
#endif

// ADV / DIF: both operators (the fused pass) or one of them (Advec_4::exec / Diff_4::exec called separately: same body, same bits)
template<class TF, int NJ, int PB, bool ADV = true, bool DIF = true>
__global__ void __launch_bounds__(64*NJ, MHH_MARCH4_OCC) rhs44_march_kernel(const GridDev<TF> g, const March4Fields<TF> f, const MarchTiling mt)
{
    constexpr int AL = (PB == 16) ? 16 / (int)sizeof(TF) : 1;
    constexpr int TI = ((70 + AL-1)/AL)*AL, TJ = NJ + 6, NT = 64*NJ, NTILE = TI*TJ;
    constexpr int RUV = 4, RW = 5;
    __shared__ __attribute__((aligned(16))) TF U[RUV][NTILE];
    __shared__ __attribute__((aligned(16))) TF V[RUV][NTILE];
    __shared__ __attribute__((aligned(16))) TF W[RW][NTILE];

    int bx, by, kcn;
    if (!decode_march(mt, blockIdx.x, bx, by, kcn)) return;
    const int jj = g.icells, kk = g.ijcells;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty*64 + tx;
    const int i0 = g.istart + bx*64, j0 = g.jstart + by*NJ;
    const int kb = g.kstart + kcn*mt.kc;
    const int ke = (kb + mt.kc < g.kend) ? kb + mt.kc : g.kend;
    const int i = i0 + tx, j = j0 + ty;
    const bool active = (i < g.iend) && (j < g.jend);
    const int ci = (i < g.iend) ? i : g.iend-1, cj = (j < g.jend) ? j : g.jend-1;
    const int col = ci + cj*jj;
    const int l = (ty+3)*TI + (tx+3);
    auto su = [](int p) { return (p + 16) % RUV; };
    auto sw = [](int p) { return (p + 20) % RW; };

    TileCopy<TF, PB, TI, TJ, NT> tc;
    tc.init(tid, i0 - 3, j0 - 3, g.icells, g.jcells);
    auto dma_tile = [&](const TF* __restrict__ fld, int kp, TF* __restrict__ lds)
    {
        if (kp < 0 || kp >= g.kcells) return;                         // wave-uniform
        tc.copy(fld + (size_t)kp*kk, lds);
    };
    auto colval = [&](const TF* __restrict__ fld, int kp) -> TF { return (kp >= 0 && kp < g.kcells) ? fld[col + kp*kk] : TF(0); };

    // ---- prologue: u, v planes kb-2..kb+1, w planes kb-1..kb+2, windows centred on kb ---------------------------------
    for (int p = kb-2; p <= kb+1; ++p) { dma_tile(f.u, p, U[su(p)]); dma_tile(f.v, p, V[su(p)]); }
    for (int p = kb-1; p <= kb+2; ++p) dma_tile(f.w, p, W[sw(p)]);
    TF uw[7], vw[7], ww[7];
#pragma unroll
    for (int n=0; n<7; ++n) { uw[n] = colval(f.u, kb-3+n); vw[n] = colval(f.v, kb-3+n); ww[n] = colval(f.w, kb-3+n); }
    // The tendencies are read one level ahead of their use (MHH_MARCH4_TPREF=0: where they are used); those of the first
    // level here, ahead of the prologue's wait, so that no load is pending when the loop is entered.
#ifndef MHH_MARCH4_TPREF
#define MHH_MARCH4_TPREF 1
#endif
    constexpr bool TPREF = (MHH_MARCH4_TPREF != 0);
    TF tnu = 0, tnv = 0, tnw = 0;
    if (TPREF && active && kb < ke) { const int c0 = col + kb*kk; tnu = stream_load(f.ut + c0); tnv = stream_load(f.vt + c0); tnw = stream_load(f.wt + c0); }
    wait_vmem();
    __syncthreads();

    const TF dxi = g.dxi_t, dyi = g.dyi_t;
    const bool dim3 = g.dim3;
    auto both = [&](TF t, const TF ad[3], const TF df[3]) -> TF
    {
        if constexpr (ADV) { t -= ad[0]; if (dim3) t -= ad[1]; t -= ad[2]; }
        if constexpr (DIF) { t += df[0]; if (dim3) t += df[1]; t += df[2]; }
        return t;
    };

    // the u and v results of level k are stored at the top of iteration k+1: the s_waitcnt vmcnt(0) in front of the
    // end-of-level barrier also waits for stores, and stores issued right before it would expose their latency
    TF ut_pending = 0, vt_pending = 0; int c_pending = -1;
    // vertical face products of the advection and inner vertical gradients of the diffusion, carried to the next level
    // (cell_ops.h, advec4_mom_vc / diff4_vc; MHH_MARCH4_CARRY=0: every level forms all four, as the cell kernels do)
#ifndef MHH_MARCH4_CARRY
#define MHH_MARCH4_CARRY 1
#endif
    constexpr bool CARRY = (MHH_MARCH4_CARRY != 0);
    TF au[3] = {0, 0, 0}, av[3] = {0, 0, 0}, aw[3] = {0, 0, 0}, du[3] = {0, 0, 0}, dv[3] = {0, 0, 0}, dw[3] = {0, 0, 0};
    const int kw0 = (kb > g.kstart) ? kb : g.kstart + 1;            // the first level of this chunk with a w equation
    for (int k = kb; k < ke; ++k)
    {
        const bool fresh = !CARRY || (k == kb), fresh_w = !CARRY || (k == kw0);
        const bool more = (k + 1 < ke);
        if (more) dma_tile(f.w, k+3, W[sw(k+3)]);
        if (c_pending >= 0) { stream_store(f.ut + c_pending, ut_pending); stream_store(f.vt + c_pending, vt_pending); c_pending = -1; }
        const TF tcu = tnu, tcv = tnv, tcw = tnw;
        if (TPREF && more && active) { const int cn = col + (k+1)*kk; tnu = stream_load(f.ut + cn); tnv = stream_load(f.vt + cn); tnw = stream_load(f.wt + cn); }
        const TF nu = more ? colval(f.u, k+4) : TF(0), nv = more ? colval(f.v, k+4) : TF(0), nw = more ? colval(f.w, k+4) : TF(0);

        const MarchView<TF, TI> Uv{{U[su(k-2)]+l, U[su(k-1)]+l, U[su(k)]+l, U[su(k+1)]+l, nullptr}, uw};
        const MarchView<TF, TI> Vv{{V[su(k-2)]+l, V[su(k-1)]+l, V[su(k)]+l, V[su(k+1)]+l, nullptr}, vw};
        const MarchView<TF, TI> Wv{{nullptr, W[sw(k-1)]+l, W[sw(k)]+l, W[sw(k+1)]+l, W[sw(k+2)]+l}, ww};
        const bool bot = (k == g.kstart), top = (k == g.kend-1);
        const int c = col + k*kk;
        TF ad[3], df[3];

        // ---- w equation first: the only reader of the u, v planes k-2 ---------------------------------------------------
        if (active && k > g.kstart)
        {
            const bool botw = (k == g.kstart+1);
            const TF gw4[4] = {uniform_load(g.dzi4, k-2), uniform_load(g.dzi4, k-1), uniform_load(g.dzi4, k), uniform_load(g.dzi4, k+1)};
            if constexpr (ADV) advec4_mom_vc<2>(ad, Wv, Uv, Vv, Wv, botw, top, dxi, dyi, uniform_load(g.dzhi4, k), dim3, aw, fresh_w);
            if constexpr (DIF) diff4_vc(df, Wv, botw, top, f.visc, g.dxidxi_t, g.dyidyi_t, gw4, uniform_load(g.dzhi4, k), dim3, dw, fresh_w);
            stream_store(f.wt + c, both(TPREF ? tcw : stream_load(f.wt + c), ad, df));
        }
        if (more)
        {
            __syncthreads();                                          // everyone is done with the u, v planes k-2
            dma_tile(f.u, k+2, U[su(k+2)]); dma_tile(f.v, k+2, V[su(k+2)]);
        }
        // ---- u and v equations (planes k of u, v; k-1..k+2 of w) --------------------------------------------------------
        if (active)
        {
            const TF gc4[4] = {uniform_load(g.dzhi4, k-1), uniform_load(g.dzhi4, k), uniform_load(g.dzhi4, k+1), uniform_load(g.dzhi4, k+2)};
            if constexpr (ADV) advec4_mom_vc<0>(ad, Uv, Uv, Vv, Wv, bot, top, dxi, dyi, uniform_load(g.dzi4, k), dim3, au, fresh);
            if constexpr (DIF) diff4_vc(df, Uv, bot, top, f.visc, g.dxidxi_d, g.dyidyi_d, gc4, uniform_load(g.dzi4, k), dim3, du, fresh);
            ut_pending = both(TPREF ? tcu : stream_load(f.ut + c), ad, df);
            if constexpr (ADV) advec4_mom_vc<1>(ad, Vv, Uv, Vv, Wv, bot, top, dxi, dyi, uniform_load(g.dzi4, k), dim3, av, fresh);
            if constexpr (DIF) diff4_vc(df, Vv, bot, top, f.visc, g.dxidxi_d, g.dyidyi_d, gc4, uniform_load(g.dzi4, k), dim3, dv, fresh);
            vt_pending = both(TPREF ? tcv : stream_load(f.vt + c), ad, df);
            c_pending = c;
        }
        wait_vmem();                  // unconditional: every path back to the loop head carries a vmcnt(0) the compiler can see
        __syncthreads();
        if (more) { shift7(uw, nu); shift7(vw, nv); shift7(ww, nw); }
    }
    if (c_pending >= 0) { stream_store(f.ut + c_pending, ut_pending); stream_store(f.vt + c_pending, vt_pending); }
}

#ifndef MHH_MARCH4_NJ
#define MHH_MARCH4_NJ 4
#endif
template<class TF>
int march4_launch(const mhh_grid* g, const mhh_fields* f, int pb, hipStream_t st, int mode = 0)      // 0: both, 1: advec_4, 2: diff_4
{
    constexpr int NJ = MHH_MARCH4_NJ;
    March4Fields<TF> mf;
    mf.u = cp<TF>(f->u); mf.v = cp<TF>(f->v); mf.w = cp<TF>(f->w);
    mf.ut = mp<TF>(f->ut); mf.vt = mp<TF>(f->vt); mf.wt = mp<TF>(f->wt); mf.visc = TF(f->visc);
    const MarchTiling t = make_march_tiling(g, NJ, MHH_MARCH4_KC);
    const dim3 nb(march_blocks(t)), bs(64, NJ);
    const GridDev<TF> gd = make_grid<TF>(g);
#define MHH_L4(PBV, A, D) hipLaunchKernelGGL((rhs44_march_kernel<TF, NJ, PBV, A, D>), nb, bs, 0, st, gd, mf, t)
    if (mode == 0)      { if (pb == 16) MHH_L4(16, true, true);  else MHH_L4(4, true, true); }
    else if (mode == 1) { if (pb == 16) MHH_L4(16, true, false); else MHH_L4(4, true, false); }
    else                { if (pb == 16) MHH_L4(16, false, true); else MHH_L4(4, false, true); }
#undef MHH_L4
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
static unsigned long long g_rhs44_march_launches = 0;
} // namespace

MHH_API unsigned long long mhh_stat_rhs44_march_launches(void) { return g_rhs44_march_launches; }

// Entry used by mhh_rhs_exec for (advec_4, diff_4): u, v, w only (scalars take the per-field kernels). Returns 1 when the
// marching kernel ran, 0 when it is switched off (MHH_RHS44_IMPL=cell), < 0 on error (-code).
static int rhs44_march_mode(const mhh_grid* g, const mhh_fields* f, void* stream, int mode);
int mhh_rhs44_march(const mhh_grid* g, const mhh_fields* f, void* stream) { return rhs44_march_mode(g, f, stream, 0); }
// Advec_4::exec / Diff_4::exec on their own for u, v, w (same return convention)
int mhh_advec4_march(const mhh_grid* g, const mhh_fields* f, void* stream) { return rhs44_march_mode(g, f, stream, 1); }
int mhh_diff4_march(const mhh_grid* g, const mhh_fields* f, void* stream) { return rhs44_march_mode(g, f, stream, 2); }
static int rhs44_march_mode(const mhh_grid* g, const mhh_fields* f, void* stream, int mode)
{
    { const char* e = getenv("MHH_RHS44_IMPL"); if (e && !strcmp(e, "cell")) return 0; }
    const int vec = (g->dtype == MHH_F64) ? 2 : 4;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    if (g->igc < 3 || g->jgc < 3 || g->kgc < 3) return 0;
    // 16-byte pieces need 16-byte aligned rows and tile origin (i0 - 3 = igc - 3 + 64*bx); other layouts copy in 4-byte pieces
    const int pb = (g->icells % vec == 0 && (g->igc - 3) % vec == 0 && al16(f->u) && al16(f->v) && al16(f->w)) ? 16 : 4;
    ++g_rhs44_march_launches;
    const int rc = (g->dtype == MHH_F64) ? march4_launch<double>(g, f, pb, as_stream(stream), mode) : march4_launch<float>(g, f, pb, as_stream(stream), mode);
    return rc == MHH_OK ? 1 : -rc;
}
