// k_march.hip -- the flagship kernel: advec_2i5 + diff_smag2 for u, v, w and one scalar in ONE pass, as a
// k-marching, LDS-staged, register-pipelined stencil sweep for gfx950.
//
// Why this shape (numbers: profiles/r1b_sweep_xcd_tiling.md): the one-thread-per-cell version issues ~180 vector
// loads per cell through the 64 B/clk L1 and is bound by that pipe, not by HBM or VALU. Here a 64 x NJ block of
// threads owns a column tile and walks up in k:
//   * the plane being updated and its two neighbours live in LDS with their halos (u, v, w: +-3; evisc: +-1),
//     loaded once per block (x1.9..2.7 halo amplification instead of x20 re-reads), read back with ds_read_b64;
//   * each thread keeps the 7-deep k-window of its own column of u, v, w, s in registers;
//   * vertical face fluxes (advective centred/upwind parts and the Smagorinsky stress) are computed once, on the
//     top face, and carried to the next level as its bottom face -- identical operands, identical rounding, so
//     the result is still bit-identical to the reference CPU path (src/advec_2i5.cxx, src/diff_smag2.cxx);
//   * the next plane travels global -> LDS by LDS-DMA (16-byte pieces on aligned rows, 4-byte pieces on any other layout)
//     into a spare ring slot while the current level is being computed: one barrier per level; a register-staged copy
//     form (two barriers) remains selectable for A/B runs (MHH_MARCH_DMA=0);
//   * fp64: the level body exists in four instantiations (interior level or not, base-state density exactly 1 or not),
//     so that on the hot path the face orders, wall predicates and density factors are compile-time constants.
// Accumulation order per tendency is the reference's: t += advec_horizontal; t += advec_vertical; t += diffusion.
// Measurements and the experiments that did not pay: profiles/r1c_march_kernel_pmc.md, profiles/r1e_kernels_pmc.md.
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include "k_common.h"
#include "k_march_common.h"
#include <gfx950_prims.h>   // angle form: the CPU emulation build (tests/emul) overrides it by include path

using namespace mhh;

// Diagnostic build only (-DMHH_MARCH_STAMPS): per-segment cycle sums of the marching loop, never compiled into the product.
#ifdef MHH_MARCH_STAMPS
__device__ unsigned long long g_march_stamps[8];
#define STAMP(n) do { const unsigned long long t_ = clock64(); stamp_acc[n] += t_ - stamp_t; stamp_t = t_; } while (0)
#else
#define STAMP(n)
#endif

namespace
{
template<class TF> __device__ __forceinline__ TF win_cen(const TF (&w)[7], int order)     // window index 3 = level k; face k+1/2
{
    if (order == 2) return i2(w[3], w[4]);
    if (order == 4) return i4ws(w[2], w[3], w[4], w[5]);
    return i6(w[1], w[2], w[3], w[4], w[5], w[6]);
}
template<class TF> __device__ __forceinline__ TF win_upw(const TF (&w)[7], int order)
{
    if (order == 4) return i3ws(w[2], w[3], w[4], w[5]);
    return i5(w[1], w[2], w[3], w[4], w[5], w[6]);
}
// vertical advective increment from the face products T = rt*w_t*I_t, B = rb*w_b*I_b, Gt = rt*|w_t|*D_t, Gb likewise
// x / rc with the wave-uniform shortcut for rc == 1 (Boussinesq base state): x / 1 is x, bit for bit, and an
// fp64 division is ~12 VALU instructions that this kernel would otherwise issue ~12 times per cell.
template<class TF> __device__ __forceinline__ TF div_rho(TF x, TF rc, bool one) { return one ? x : x / rc; }

template<class TF> __device__ __forceinline__ TF vert_combine(int ot, int ob, TF T, TF B, TF Gt, TF Gb, TF rc, bool one, TF dz)
{
    TF cen;
    if (ob == 0)      cen = - div_rho( T, rc, one ) * dz;
    else if (ot == 0) cen = - div_rho( -B, rc, one ) * dz;
    else              cen = - div_rho( T - B, rc, one ) * dz;
    const bool ut = (ot >= 4), ub = (ob >= 4);
    if (ut && ub) return cen + div_rho( Gt - Gb, rc, one ) * dz;
    if (ut)       return cen + div_rho( Gt, rc, one ) * dz;
    if (ub)       return cen - div_rho( Gb, rc, one ) * dz;
    return cen;
}
template<class TF> __device__ __forceinline__ void shift(TF (&w)[7], TF nw)
{
    w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = w[4]; w[4] = w[5]; w[5] = w[6]; w[6] = nw;
}

template<class TF> struct MarchFields
{
    const TF* __restrict__ u; const TF* __restrict__ v; const TF* __restrict__ w; const TF* __restrict__ s; const TF* __restrict__ ev;
    TF* __restrict__ ut; TF* __restrict__ vt; TF* __restrict__ wt; TF* __restrict__ st;
    const TF* __restrict__ rhoref; const TF* __restrict__ rhorefh;
    const TF* __restrict__ ufb; const TF* __restrict__ uft; const TF* __restrict__ vfb; const TF* __restrict__ vft;
    const TF* __restrict__ sfb; const TF* __restrict__ sft;
    TF visc, svisc, tPr; int sm;
    const TF* __restrict__ threfh; TF grav;      // folded dry buoyancy of the scalar (threfh == nullptr: off)
};

#ifndef MHH_MARCH_OCC
#define MHH_MARCH_OCC 2           // fp64: 213-237 VGPRs, 74.6 KB LDS -> two blocks per CU
#endif
#ifndef MHH_MARCH_OCC_F32
#define MHH_MARCH_OCC_F32 4       // fp32: half the registers and LDS -> four waves per SIMD (gabls1 1024x1024x256: 8.5 -> 7.9 ms)
#endif
// DMA = true : planes travel global -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write, one
//              barrier per level; rings one slot deeper so that the copy of the next plane can run under the whole
//              compute phase). Needs 16-byte aligned rows: icells % (16/sizeof(TF)) == 0 and 16-byte aligned fields.
// DMA = false: planes are staged through registers (prefetch, two barriers per level): any alignment.
// PB = 16 : LDS-DMA in 16-byte pieces (rows 16-byte aligned); PB = 4: LDS-DMA in 4-byte pieces (global_load_lds_dword):
//           any layout, four times the copy instructions; PB = 0: register-staged.
// ADV / DIF: which operator's terms are added -- both (the fused pass), or one of them: Advec::exec and Diff::exec as
// separate calls then run the same kernel body (same bits, same order of accumulation as the fused pass in two steps).
template<class TF, int NJ, bool HAS_S, int PB, bool ADV = true, bool DIF = true>
__global__ void __launch_bounds__(64*NJ, (sizeof(TF) == 4 ? MHH_MARCH_OCC_F32 : MHH_MARCH_OCC)) rhs25_march_kernel(const GridDev<TF> g, const MarchFields<TF> f, const MarchTiling mt)
{
    constexpr bool DMA = (PB != 0);
    constexpr int VEC = 16 / (int)sizeof(TF);                       // elements per 16-byte DMA piece
    constexpr int AL = (PB == 16) ? VEC : 1;                        // granularity of tile widths / origins in elements
    constexpr int TI = ((70 + AL-1)/AL)*AL;                         // u,v,w,s tile: x from i0-3
    constexpr int EX = (PB == 16) ? VEC : 1;                        // evisc tile: x from i0-EX (aligned for 16-byte DMA)
    constexpr int TE = ((64 + EX + 1 + AL-1)/AL)*AL;
    constexpr int TJ = NJ + 6, TJE = NJ + 2, NT = 64*NJ;
    constexpr int NTILE = TI*TJ, NETILE = TE*TJE;
    constexpr int RU = DMA ? 3 : 2, RW = DMA ? 3 : 2, RE = DMA ? 4 : 3, RS = DMA ? 2 : 1;
    // LDS rings: only the planes a level reads with horizontal offsets are kept -- u, v: k-1 and k (their k+1 values
    // are needed at the thread's own column only: register window); w: k and k+1; evisc: k-1..k+1; scalar: k.
    __shared__ __attribute__((aligned(16))) TF U[RU][NTILE];
    __shared__ __attribute__((aligned(16))) TF V[RU][NTILE];
    __shared__ __attribute__((aligned(16))) TF W[RW][NTILE];
    __shared__ __attribute__((aligned(16))) TF S[HAS_S ? RS : 1][HAS_S ? NTILE : VEC];
    __shared__ __attribute__((aligned(16))) TF E[DIF ? RE : 1][DIF ? NETILE : VEC];

    int bx, by, kcn;
    if (!decode_march(mt, blockIdx.x, bx, by, kcn)) return;        // whole block leaves together: no barrier hazard
    const int jj = g.icells, kk = g.ijcells;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty*64 + tx;
    const int i0 = g.istart + bx*64, j0 = mt.jbase + by*NJ;
    const int kb = g.kstart + kcn*mt.kc;
    const int ke = (kb + mt.kc < g.kend) ? kb + mt.kc : g.kend;
    const int i = i0 + tx, j = j0 + ty;
    const bool active = (i < g.iend) && (j < mt.jlim);
    const int ci = (i < g.iend) ? i : g.iend-1, cj = (j < mt.jlim) ? j : mt.jlim-1;   // clamped column for the window loads
    const int col = ci + cj*jj;
    const int ij = col;
    // tendencies are touched once per kernel: their loads and stores carry the non-temporal hint, so that they stream past L2
    // instead of evicting the field planes that neighbouring tiles re-read (512^3: HBM fetch 15.4 -> 14.1 GB per launch, same
    // time; -DMHH_MARCH_NO_NT for A/B runs)
#ifndef MHH_MARCH_NO_NT
    auto tld = [](const TF* q) -> TF { return stream_load(q); };
    auto tst = [](TF* q, TF v) { stream_store(q, v); };
#else
    auto tld = [](const TF* q) -> TF { return *q; };
    auto tst = [](TF* q, TF v) { *q = v; };
#endif
    const int l = (ty+3)*TI + (tx+3), le = (ty+1)*TE + (tx+EX);
    auto slot = [](int p, int r) { return (p + 12) % r; };            // 12 is a multiple of every ring depth

    // ---- tile movers. A tile is walked in pieces of PW 32-bit words: e = tid + n*NT; piece -> (row, first word) --------
    constexpr int EW = (int)sizeof(TF) / 4;                           // words per element
    constexpr int PW = DMA ? PB/4 : EW;                               // words per piece (staged: one element)
    constexpr int PPR = TI*EW / PW, PPRE = TE*EW / PW;                // pieces per tile row
    constexpr int NP = PPR*TJ, NPE = PPRE*TJE;
    constexpr int NLD = (NP + NT - 1) / NT, NLDE = (NPE + NT - 1) / NT;
    int off[NLD], offe[NLDE];                                         // word offsets from the start of a plane
    bool okt[NLD], oke[NLDE];
#pragma unroll
    for (int n=0; n<NLD; ++n)
    {
        const int e = tid + n*NT;
        const int tj = e / PPR, tw = (e - tj*PPR)*PW;
        const int gw = (i0 - 3)*EW + tw, gj = j0 - 3 + tj;
        okt[n] = (e < NP) && (gw + PW <= g.icells*EW) && (gj < g.jcells);
        off[n] = okt[n] ? gw + gj*jj*EW : 0;
    }
#pragma unroll
    for (int n=0; n<NLDE; ++n)
    {
        const int e = tid + n*NT;
        const int tj = e / PPRE, tw = (e - tj*PPRE)*PW;
        const int gw = (i0 - EX)*EW + tw, gj = j0 - 1 + tj;
        oke[n] = (e < NPE) && (gw + PW <= g.icells*EW) && (gj < g.jcells);
        offe[n] = oke[n] ? gw + gj*jj*EW : 0;
    }
    // register-staged movers
    auto ld_tile = [&](const TF* __restrict__ fld, int kp, TF (&r)[NLD])
    {
        const bool kok = (kp >= 0) && (kp < g.kcells);
        const TF* __restrict__ pl = fld + (kok ? (size_t)kp*kk : 0);
#pragma unroll
        for (int n=0; n<NLD; ++n) r[n] = (kok && okt[n]) ? pl[off[n]/EW] : TF(0);
    };
    auto st_tile = [&](TF* __restrict__ lds, const TF (&r)[NLD])
    {
#pragma unroll
        for (int n=0; n<NLD; ++n) { const int e = tid + n*NT; if (n+1 < NLD || e < NP) lds[e] = r[n]; }
    };
    auto ld_etile = [&](int kp, TF (&r)[NLDE])
    {
        const bool kok = (kp >= 0) && (kp < g.kcells);
        const TF* __restrict__ pl = f.ev + (kok ? (size_t)kp*kk : 0);
#pragma unroll
        for (int n=0; n<NLDE; ++n) r[n] = (kok && oke[n]) ? pl[offe[n]/EW] : TF(0);
    };
    auto st_etile = [&](TF* __restrict__ lds, const TF (&r)[NLDE])
    {
#pragma unroll
        for (int n=0; n<NLDE; ++n) { const int e = tid + n*NT; if (n+1 < NLDE || e < NPE) lds[e] = r[n]; }
    };
    // LDS-DMA movers: PB bytes per lane straight into the ring slot; lanes outside the tile / the array sit out
    const int wave_e0 = tid & ~63;                                    // first piece index of this wave within a sweep
    // SV: the copy as scalar base + per-lane byte offset + scalar LDS address (gfx950_prims.h): no vector ALU per piece
#ifdef MHH_DMA_NO_SV
    constexpr bool SV = false;
#else
    constexpr bool SV = DMA && (MHH_RAW_DMA != 0);
#endif
    const unsigned wave_lds = uniform_u32((unsigned)(wave_e0*PW*4));
    auto dma_piece = [&](const TF* plane, int word_off, TF* lds, int n)
    {
        if constexpr (SV) lds_dma_sv<PB>(plane, (unsigned)word_off*4u, lds_address(lds) + (wave_lds + (unsigned)(n*NT*PW*4)));
        else
        {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(plane) + word_off;
            uint32_t* dst = reinterpret_cast<uint32_t*>(lds) + (size_t)(wave_e0 + n*NT)*PW;
            if constexpr (PB == 16) lds_dma16<(sizeof(TF) == 8)>(src, dst); else lds_dma4<(sizeof(TF) == 8)>(src, dst);
        }
    };
    auto dma_tile = [&](const TF* __restrict__ fld, int kp, TF* __restrict__ lds)
    {
        if (kp < 0 || kp >= g.kcells) return;                         // wave-uniform
        const TF* __restrict__ pl = fld + (size_t)kp*kk;
#pragma unroll
        for (int n=0; n<NLD; ++n)
            if (okt[n]) dma_piece(pl, off[n], lds, n);
    };
    auto dma_etile = [&](int kp, TF* __restrict__ lds)
    {
        if (kp < 0 || kp >= g.kcells) return;
        const TF* __restrict__ pl = f.ev + (size_t)kp*kk;
#pragma unroll
        for (int n=0; n<NLDE; ++n)
            if (oke[n]) dma_piece(pl, offe[n], lds, n);
    };
    auto colval = [&](const TF* __restrict__ fld, int kp) -> TF
    {
        return (kp >= 0 && kp < g.kcells) ? fld[col + kp*kk] : TF(0);
    };

    // ---- prologue: the planes level ks reads, scalar plane ks, windows centred on ks ------------------------------
    const int ks = kb - 1;                 // warm-up level: only top-face quantities are formed there
    if constexpr (DMA)
    {
        for (int p = ks-1; p <= ks+1; ++p)
        {
            if (p <= ks) { dma_tile(f.u, p, U[slot(p, RU)]); dma_tile(f.v, p, V[slot(p, RU)]); }
            if (p >= ks) dma_tile(f.w, p, W[slot(p, RW)]);
            if (DIF) dma_etile(p, E[slot(p, RE)]);
        }
        if (HAS_S) dma_tile(f.s, ks, S[slot(ks, RS)]);
    }
    else
    {
        TF r[NLD]; TF re[NLDE];
        for (int p = ks-1; p <= ks+1; ++p)
        {
            if (p <= ks) { ld_tile(f.u, p, r); st_tile(U[slot(p, RU)], r); ld_tile(f.v, p, r); st_tile(V[slot(p, RU)], r); }
            if (p >= ks) { ld_tile(f.w, p, r); st_tile(W[slot(p, RW)], r); }
            if (DIF) { ld_etile(p, re);    st_etile(E[slot(p, RE)], re); }
        }
        if (HAS_S) { ld_tile(f.s, ks, r); st_tile(S[0], r); }
    }
    TF uw[7], vw[7], ww[7], sw[7];
#pragma unroll
    for (int n=0; n<7; ++n)
    {
        uw[n] = colval(f.u, ks-3+n); vw[n] = colval(f.v, ks-3+n); ww[n] = colval(f.w, ks-3+n);
        sw[n] = HAS_S ? colval(f.s, ks-3+n) : TF(0);
    }
    if constexpr (DMA) wait_vmem();
    __syncthreads();

    // carried bottom-face products: advective centred (T) and upwind (G) parts, diffusive flux (D)
    TF cTu = 0, cGu = 0, cDu = 0, cTv = 0, cGv = 0, cDv = 0, cTw = 0, cGw = 0, cDw = 0, cTs = 0, cGs = 0, cDs = 0;

    // Latency of the tendency read-modify-writes (fp64 form; the fp32 form has no registers to spare at four waves per SIMD):
    //  * TPREF: the tendencies of the NEXT level are loaded a whole level ahead of their use, like the LDS-DMA planes;
    //  * DSTORE: the last tendency finished in a level (the scalar's) is stored at the top of the next level, so that the
    //    s_waitcnt vmcnt(0) in front of the barrier -- which on gfx9 also waits for stores -- does not find it just issued.
    //    (1: all four deferred -- spills; 2: w and scalar; 3: scalar only.)
    // 512^3: 6.05 -> 5.75 (TPREF) -> 5.67 ms (DSTORE 3), 225 -> 245 VGPRs, still two waves per SIMD and no scratch.
#ifndef MHH_MARCH_TPREF
#define MHH_MARCH_TPREF 1
#endif
#ifndef MHH_MARCH_DSTORE
#define MHH_MARCH_DSTORE 3
#endif
    constexpr bool TPREF = (sizeof(TF) == 8) && (MHH_MARCH_TPREF != 0);
    constexpr int DSTORE = (sizeof(TF) == 8) ? MHH_MARCH_DSTORE : 0;       // 0: every store where its value is finished
    TF dsu = 0, dsv = 0, dsw = 0, dss = 0; int dsk = -1; bool dsw_on = false;
    TF tpu = 0, tpv = 0, tpw = 0, tps = 0;
    const TF dxi = g.dxi_t, dyi = g.dyi_t;          // advection spelling TF(1.)/dx
    const TF dxd = g.dxi_d, dyd = g.dyi_d;          // diffusion spelling TF(1./dx)
    const TF visc = f.visc;

#ifdef MHH_MARCH_STAMPS
    unsigned long long stamp_acc[8] = {0,0,0,0,0,0,0,0}; unsigned long long stamp_t = clock64();
#endif
    // One level. FAST = an interior level of an updating iteration: every vertical face is 6th/5th order, no wall or
    // surface-flux branch applies -- the face orders and the wall predicates become constants and their dispatch
    // (a third of the loop's scalar / control instructions) disappears. Same arithmetic either way.
    // RHO1 = rhoref and rhorefh are exactly 1 on every level this block touches (Boussinesq base state): 1*x and x/1 are x,
    // bit for bit, so the density factors and divisions drop out of the instantiation instead of being tested per use.
    auto level = [&](const int k, auto fast_tag, auto rho1_tag)
    {
        constexpr bool FAST = decltype(fast_tag)::value, RHO1 = decltype(rho1_tag)::value;
        auto R = [](TF r, TF x) { return RHO1 ? x : r*x; };
        STAMP(0);
        // ---- start moving the next level's planes: k+1 of u, v, s; k+2 of w, evisc; window value k+4 ---------------
        TF pu[NLD], pv[NLD], pw[NLD], ps[NLD], pe[NLDE];        // staging registers (unused, and removed, in the DMA variant)
        const bool more = (k + 1 < ke);
        if (more)
        {
            if constexpr (DMA)
            {
                dma_tile(f.u, k+1, U[slot(k+1, RU)]); dma_tile(f.v, k+1, V[slot(k+1, RU)]); dma_tile(f.w, k+2, W[slot(k+2, RW)]);
                if (DIF) dma_etile(k+2, E[slot(k+2, RE)]);
                if (HAS_S) dma_tile(f.s, k+1, S[slot(k+1, RS)]);
            }
            else
            {
                ld_tile(f.u, k+1, pu); ld_tile(f.v, k+1, pv); ld_tile(f.w, k+2, pw); if (DIF) ld_etile(k+2, pe);
                if (HAS_S) ld_tile(f.s, k+1, ps);
            }
        }
        const TF nu = more ? colval(f.u, k+4) : TF(0), nv = more ? colval(f.v, k+4) : TF(0), nw = more ? colval(f.w, k+4) : TF(0);
        const TF ns = (more && HAS_S) ? colval(f.s, k+4) : TF(0);
        if (DSTORE && dsk >= 0 && active)
        {
            const int cd = col + dsk*kk;
            if (DSTORE == 1) { tst(f.ut + cd, dsu); tst(f.vt + cd, dsv); }
            if (DSTORE <= 2 && dsw_on) tst(f.wt + cd, dsw);
            if (HAS_S) tst(f.st + cd, dss);
        }
        dsk = -1;
        const TF tcu = tpu, tcv = tpv, tcw = tpw, tcs = tps;      // this level's tendencies (loaded during the previous level)
        if (TPREF && more && active) {     // the warm-up level ks = kb-1 fetches those of kb
            const int cn = col + (k+1)*kk; tpu = tld(f.ut + cn); tpv = tld(f.vt + cn); tpw = tld(f.wt + cn); if (HAS_S) tps = tld(f.st + cn); }

        const TF* __restrict__ uk = U[slot(k, RU)] + l;  const TF* __restrict__ ukm = U[slot(k-1, RU)] + l;
        const TF* __restrict__ vk = V[slot(k, RU)] + l;  const TF* __restrict__ vkm = V[slot(k-1, RU)] + l;
        const TF* __restrict__ wk = W[slot(k, RW)] + l;  const TF* __restrict__ wkp = W[slot(k+1, RW)] + l;
        const TF* __restrict__ sk = S[HAS_S ? slot(k, RS) : 0] + (HAS_S ? l : 0);
        const TF* __restrict__ ek = E[DIF ? slot(k, RE) : 0] + (DIF ? le : 0); const TF* __restrict__ ekm = E[DIF ? slot(k-1, RE) : 0] + (DIF ? le : 0);
        const TF* __restrict__ ekp = E[DIF ? slot(k+1, RE) : 0] + (DIF ? le : 0);
        // per-level coefficients (wave-uniform)
        // per-level coefficients: scalar loads (uniform_load), not vector loads whose wait would drain the copies in flight
        const TF rhkp = RHO1 ? TF(1) : uniform_load(f.rhorefh, k+1), rhk = RHO1 ? TF(1) : uniform_load(f.rhorefh, k), rk = RHO1 ? TF(1) : uniform_load(f.rhoref, k);
        const TF dzi = uniform_load(g.dzi, k), dzhi = uniform_load(g.dzhi, k), dzhip = uniform_load(g.dzhi, k+1);
        const bool rk1 = RHO1 || (rk == TF(1.)), rhk1 = RHO1 || (rhk == TF(1.));
        const int otc = FAST ? 6 : order_face_c(k+1, g.kstart, g.kend);
        const int obc = FAST ? 6 : order_face_c(k, g.kstart, g.kend);
        const bool wlev = FAST || (k >= g.kstart);               // the w equation's "faces" are cell centres kstart..kend-1
        const int otw = FAST ? 6 : (wlev ? order_face_w(k, g.kstart, g.kend) : 0);
        const int obw = FAST ? 6 : ((k-1 >= g.kstart) ? order_face_w(k-1, g.kstart, g.kend) : 0);
        // surface model: the lowest / highest level takes the prescribed flux instead of the resolved one
        const bool fb = !FAST && f.sm && (k == g.kstart), ft = !FAST && f.sm && (k == g.kend-1);
        const bool need_dtop = FAST || (!(ft) && (k < g.kend-1 || !f.sm) && (k+1 <= g.kend));   // top diffusive flux of level k is used by k or k+1

        STAMP(1);
        // ---- top-face quantities of level k --------------------------------------------------------------------
        TF Tu = 0, Gu = 0, Tv = 0, Gv = 0, Tw = 0, Gw = 0, Ts = 0, Gs = 0;
        if (ADV && otc != 0)
        {
            const TF wtu = i2(wkp[-1], wkp[0]);
            const TF wtv = i2(wkp[-TI], wkp[0]);
            Tu = R(rhkp, wtu) * win_cen(uw, otc);
            Tv = R(rhkp, wtv) * win_cen(vw, otc);
            if (otc >= 4) { Gu = R(rhkp, tabs(wtu)) * win_upw(uw, otc); Gv = R(rhkp, tabs(wtv)) * win_upw(vw, otc); }
            if (HAS_S)
            {
                Ts = R(rhkp, ww[4]) * win_cen(sw, otc);
                if (otc >= 4) Gs = R(rhkp, tabs(ww[4])) * win_upw(sw, otc);
            }
        }
        if (ADV && wlev)
        {
            const TF wtw = i2(ww[3], ww[4]);
            Tw = R(rk, wtw) * win_cen(ww, otw);
            if (otw >= 4) Gw = R(rk, tabs(wtw)) * win_upw(ww, otw);
        }
        TF Du = 0, Dv = 0, Dw = 0, Ds = 0;
        if (DIF && need_dtop)
        {
            const TF etu = TF(0.25)*(ek[-1] + ek[0] + ekp[-1] + ekp[0]) + visc;
            Du = R(rhkp, etu)*((uw[4]-uw[3])*dzhip + (wkp[0]-wkp[-1])*dxd);
            const TF etv = TF(0.25)*(ek[-TE] + ek[0] + ekp[-TE] + ekp[0]) + visc;
            Dv = R(rhkp, etv)*((vw[4]-vw[3])*dzhip + (wkp[0]-wkp[-TI])*dyd);
            if (HAS_S)
            {
                const TF ets = TF(0.5)*(ek[0]+ekp[0])/f.tPr + f.svisc;
                Ds = R(rhkp, ets)*(sw[4]-sw[3])*dzhip;
            }
        }
        if (DIF && wlev)
        {
            const TF etw = ek[0] + visc;
            Dw = R(rk, etw)*(ww[4]-ww[3])*dzi;
        }

        STAMP(2);
        // ---- update the tendencies of level k ----------------------------------------------------------------
        if ((FAST || k >= kb) && active)
        {
            const int c = col + k*kk;
            {   // u
                TF ue = 0, uwf = 0, vn = 0, vs = 0;
                if constexpr (ADV) { ue = i2(uk[0], uk[1]); uwf = i2(uk[-1], uk[0]);
                                     vn = i2(vk[TI-1], vk[TI]); vs = i2(vk[-1], vk[0]); }
                TF t = TPREF ? tcu : tld(f.ut + c);
                if constexpr (ADV)
                {
                    t += advec25_hor(uk, 0, TI, ue, uwf, vn, vs, dxi, dyi);
                    t += vert_combine(otc, obc, Tu, cTu, Gu, cGu, rk, rk1, dzi);
                }
                if constexpr (DIF)
                {
                    const TF ee = ek[0] + visc, ew = ek[-1] + visc;
                    const TF en = TF(0.25)*(ek[-1   ] + ek[0  ] + ek[-1+TE] + ek[TE]) + visc;
                    const TF es = TF(0.25)*(ek[-1-TE] + ek[-TE] + ek[-1   ] + ek[0 ]) + visc;
                    const TF hor = + ( ee*(uk[1]-uk[0])*dxd - ew*(uk[0]-uk[-1])*dxd ) * TF(2.)*dxd
                                   + ( en*((uk[TI]-uk[0  ])*dyd + (vk[TI]-vk[TI-1])*dxd)
                                     - es*((uk[0 ]-uk[-TI])*dyd + (vk[0 ]-vk[-1  ])*dxd) ) * dyd;
                    TF ver;
                    if (fb)      ver = div_rho( Du + rhk * f.ufb[ij], rk, rk1 ) * dzi;
                    else if (ft) ver = div_rho( - rhkp * f.uft[ij] - cDu, rk, rk1 ) * dzi;
                    else         ver = div_rho( Du - cDu, rk, rk1 ) * dzi;
                    t += hor + ver;
                }
                dsk = k;
                if (DSTORE == 1) dsu = t; else tst(f.ut + c, t);
            }
            {   // v
                TF ue = 0, uwf = 0, vn = 0, vs = 0;
                if constexpr (ADV) { ue = i2(uk[1-TI], uk[1]); uwf = i2(uk[-TI], uk[0]);
                                     vn = i2(vk[0], vk[TI]); vs = i2(vk[-TI], vk[0]); }
                TF t = TPREF ? tcv : tld(f.vt + c);
                if constexpr (ADV)
                {
                    t += advec25_hor(vk, 0, TI, ue, uwf, vn, vs, dxi, dyi);
                    t += vert_combine(otc, obc, Tv, cTv, Gv, cGv, rk, rk1, dzi);
                }
                if constexpr (DIF)
                {
                    const TF ee = TF(0.25)*(ek[-TE  ] + ek[0 ] + ek[1-TE] + ek[1]) + visc;
                    const TF ew = TF(0.25)*(ek[-1-TE] + ek[-1] + ek[-TE ] + ek[0]) + visc;
                    const TF en = ek[0] + visc, es = ek[-TE] + visc;
                    const TF hor = + ( ee*((vk[1]-vk[0 ])*dxd + (uk[1]-uk[1-TI])*dyd)
                                     - ew*((vk[0]-vk[-1])*dxd + (uk[0]-uk[-TI ])*dyd) ) * dxd
                                   + ( en*(vk[TI]-vk[0])*dyd - es*(vk[0]-vk[-TI])*dyd ) * TF(2.)*dyd;
                    TF ver;
                    if (fb)      ver = div_rho( Dv + rhk * f.vfb[ij], rk, rk1 ) * dzi;
                    else if (ft) ver = div_rho( - rhkp * f.vft[ij] - cDv, rk, rk1 ) * dzi;
                    else         ver = div_rho( Dv - cDv, rk, rk1 ) * dzi;
                    t += hor + ver;
                }
                if (DSTORE == 1) dsv = t; else tst(f.vt + c, t);
            }
            dsw_on = (FAST || k > g.kstart);
            if (FAST || k > g.kstart)
            {   // w
                TF ue = 0, uwf = 0, vn = 0, vs = 0;
                if constexpr (ADV) { ue = i2(ukm[1], uk[1]); uwf = i2(ukm[0], uk[0]);
                                     vn = i2(vkm[TI], vk[TI]); vs = i2(vkm[0], vk[0]); }
                TF t = TPREF ? tcw : tld(f.wt + c);
                if (HAS_S && f.threfh) { const TF th_k = uniform_load(f.threfh, k); t += f.grav/th_k * (i2(sw[2], sw[3]) - th_k); }   // src/thermo_dry.cxx:165-178
                if constexpr (ADV)
                {
                    t += advec25_hor(wk, 0, TI, ue, uwf, vn, vs, dxi, dyi);
                    t += vert_combine(otw, obw, Tw, cTw, Gw, cGw, rhk, rhk1, dzhi);
                }
                if constexpr (DIF)
                {
                    const TF ee = TF(0.25)*(ekm[0  ] + ek[0  ] + ekm[1 ] + ek[1 ]) + visc;
                    const TF ew = TF(0.25)*(ekm[-1 ] + ek[-1 ] + ekm[0 ] + ek[0 ]) + visc;
                    const TF en = TF(0.25)*(ekm[0  ] + ek[0  ] + ekm[TE] + ek[TE]) + visc;
                    const TF es = TF(0.25)*(ekm[-TE] + ek[-TE] + ekm[0 ] + ek[0 ]) + visc;
                    t += + ( ee*((wk[1 ]-wk[0  ])*dxd + (uk[1 ]-ukm[1 ])*dzhi)
                           - ew*((wk[0 ]-wk[-1 ])*dxd + (uk[0 ]-ukm[0 ])*dzhi) ) * dxd
                         + ( en*((wk[TI]-wk[0  ])*dyd + (vk[TI]-vkm[TI])*dzhi)
                           - es*((wk[0 ]-wk[-TI])*dyd + (vk[0 ]-vkm[0 ])*dzhi) ) * dyd
                         + div_rho( Dw - cDw, rhk, rhk1 ) * TF(2.)*dzhi;
                }
                if (DSTORE == 1 || DSTORE == 2) dsw = t; else tst(f.wt + c, t);
            }
            if (HAS_S)
            {   // scalar
                TF t = TPREF ? tcs : tld(f.st + c);
                if constexpr (ADV)
                {
                    t += advec25_hor(sk, 0, TI, uk[1], uk[0], vk[TI], vk[0], dxi, dyi);
                    t += vert_combine(otc, obc, Ts, cTs, Gs, cGs, rk, rk1, dzi);
                }
                if constexpr (DIF)
                {
                    const TF ee = TF(0.5)*(ek[0  ]+ek[1 ])/f.tPr + f.svisc;
                    const TF ew = TF(0.5)*(ek[-1 ]+ek[0 ])/f.tPr + f.svisc;
                    const TF en = TF(0.5)*(ek[0  ]+ek[TE])/f.tPr + f.svisc;
                    const TF es = TF(0.5)*(ek[-TE]+ek[0 ])/f.tPr + f.svisc;
                    const TF hor = + ( ee*(sk[1 ]-sk[0]) - ew*(sk[0]-sk[-1 ]) ) * g.dxidxi_d
                                   + ( en*(sk[TI]-sk[0]) - es*(sk[0]-sk[-TI]) ) * g.dyidyi_d;
                    TF ver;
                    if (fb)      ver = div_rho( Ds + rhk * f.sfb[ij], rk, rk1 ) * dzi;
                    else if (ft) ver = div_rho( -rhkp * f.sft[ij] - cDs, rk, rk1 ) * dzi;
                    else         ver = div_rho( Ds - cDs, rk, rk1 ) * dzi;
                    t += hor + ver;
                }
                if (DSTORE) dss = t; else tst(f.st + c, t);
            }
        }
        STAMP(3);
        // ---- carry the top faces down, rotate the rings, shift the windows --------------------------------------------
        cTu = Tu; cGu = Gu; cDu = Du; cTv = Tv; cGv = Gv; cDv = Dv; cTw = Tw; cGw = Gw; cDw = Dw; cTs = Ts; cGs = Gs; cDs = Ds;
        if constexpr (DMA)
        {
            // unconditional (also after the chunk's last level, where nothing is in flight): every path back to the loop head
            // then carries a vmcnt(0) the compiler can see, and it inserts no wait of its own in the next level
            wait_vmem();                                           // this wave's copies have landed (and its stores have left)
            STAMP(4);
            __syncthreads();                                       // ... everyone's have, and everyone is done with the oldest planes
            STAMP(6);
        }
        if (more)
        {
            if constexpr (!DMA)
            {
                __syncthreads();                                   // everyone is done reading the planes that are about to be replaced
                STAMP(4);
                st_tile(U[slot(k+1, RU)], pu); st_tile(V[slot(k+1, RU)], pv); st_tile(W[slot(k+2, RW)], pw); if (DIF) st_etile(E[slot(k+2, RE)], pe);
                if (HAS_S) st_tile(S[0], ps);
                STAMP(5);
                __syncthreads();
                STAMP(6);
            }
            shift(uw, nu); shift(vw, nv); shift(ww, nw); shift(sw, ns);
        }
    };
    bool rho_one = true;                                           // wave-uniform: scalar loads of the chunk's base state
    for (int k = ks; k <= ke; ++k) rho_one = rho_one && (f.rhoref[k] == TF(1.)) && (f.rhorefh[k] == TF(1.));
#ifdef MHH_MARCH_NO_RHO1   // A/B builds only
    rho_one = false;
#endif
    for (int k = ks; k < ke; ++k)
    {
        // interior level of an updating iteration: faces k and k+1 of the centred fields and the w "faces" k-1, k all 6th order
#ifdef MHH_MARCH_NO_FAST    // A/B builds only
        const bool fast = false;
#else
        const bool fast = (k >= kb) && (k >= g.kstart+3) && (k <= g.kend-4);
#endif
        if constexpr (sizeof(TF) == 4)
            level(k, std::false_type{}, std::false_type{});       // fp32: one body (127 VGPRs, 3-4 waves per SIMD); four bodies cost it a wave
        else if (rho_one) { if (fast) level(k, std::true_type{}, std::true_type{});  else level(k, std::false_type{}, std::true_type{}); }
        else              { if (fast) level(k, std::true_type{}, std::false_type{}); else level(k, std::false_type{}, std::false_type{}); }
    }
    if (DSTORE && dsk >= 0 && active)
    {
        const int cd = col + dsk*kk;
        if (DSTORE == 1) { tst(f.ut + cd, dsu); tst(f.vt + cd, dsv); }
        if (DSTORE <= 2 && dsw_on) tst(f.wt + cd, dsw);
        if (HAS_S) tst(f.st + cd, dss);
    }
#ifdef MHH_MARCH_STAMPS
    if ((threadIdx.x & 63) == 0) for (int n=0; n<8; ++n) atomicAdd(&g_march_stamps[n], stamp_acc[n]);
#endif
}

// mode 0: advec_2i5 + diff_smag2 (the fused pass); 1: advec_2i5 only (p may be null); 2: diff_smag2 only
template<class TF>
int march_launch(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, hipStream_t st, int mode = 0)
{
#ifndef MHH_MARCH_NJ
#define MHH_MARCH_NJ 4
#endif
    constexpr int NJ = MHH_MARCH_NJ;
    const GridDev<TF> gd = make_grid<TF>(g);
    MarchFields<TF> mf;
    mf.u = cp<TF>(f->u); mf.v = cp<TF>(f->v); mf.w = cp<TF>(f->w); mf.ev = (mode == 1) ? nullptr : cp<TF>(f->evisc);
    mf.ut = mp<TF>(f->ut); mf.vt = mp<TF>(f->vt); mf.wt = mp<TF>(f->wt);
    const bool has_s = f->nscalars >= 1;
    mf.s = has_s ? cp<TF>(f->s[0]) : nullptr; mf.st = has_s ? mp<TF>(f->st[0]) : nullptr;
    mf.rhoref = cp<TF>(f->rhoref); mf.rhorefh = cp<TF>(f->rhorefh);
    mf.ufb = cp<TF>(f->u_fluxbot); mf.uft = cp<TF>(f->u_fluxtop); mf.vfb = cp<TF>(f->v_fluxbot); mf.vft = cp<TF>(f->v_fluxtop);
    mf.sfb = has_s ? cp<TF>(f->s_fluxbot[0]) : nullptr; mf.sft = has_s ? cp<TF>(f->s_fluxtop[0]) : nullptr;
    mf.visc = TF(f->visc); mf.svisc = has_s ? TF(f->svisc[0]) : TF(0); mf.tPr = p ? TF(p->tPr) : TF(1); mf.sm = (p && mode != 1) ? p->surface_model : 0;
    const bool buoy = mode == 0 && has_s && p->buoyancy == 2 && p->th_for_N2 == 0;
    mf.threfh = buoy ? cp<TF>(p->threfh) : nullptr; mf.grav = buoy ? TF(p->grav) : TF(0);
#ifndef MHH_MARCH_KC
#define MHH_MARCH_KC 128
#endif
    // a strip of a few rows (mhh_rhs_exec_rows on the edge rows) takes short k-chunks: enough blocks to fill the GPU
    const int kc = (j0 >= 0 && (j1 - j0) * 4 <= g->jmax) ? 16 : MHH_MARCH_KC;
    const MarchTiling t = make_march_tiling(g, NJ, kc, j0, j1);
    const unsigned nblocks = march_blocks(t);
    // 16-byte LDS-DMA needs 16-byte aligned plane rows; other layouts copy in 4-byte pieces (MHH_MARCH_DMA=4 forces that
    // form, =0 the register-staged one; same arithmetic in all three)
    constexpr int VEC = 16 / (int)sizeof(TF);
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    const char* env = getenv("MHH_MARCH_DMA");
    const bool aligned = (g->icells % VEC == 0) && al16(f->u) && al16(f->v) && al16(f->w) && (mode == 1 || al16(f->evisc)) && (!has_s || al16(f->s[0]));
    const int pb = (env && !strcmp(env, "0")) ? 0 : ((env && !strcmp(env, "4")) || !aligned) ? 4 : 16;
#define MHH_LAUNCH_MARCH(PBV) do { \
        if (has_s) hipLaunchKernelGGL((rhs25_march_kernel<TF, NJ, true, PBV>),  dim3(nblocks), dim3(64, NJ), 0, st, gd, mf, t); \
        else       hipLaunchKernelGGL((rhs25_march_kernel<TF, NJ, false, PBV>), dim3(nblocks), dim3(64, NJ), 0, st, gd, mf, t); } while (0)
#define MHH_LAUNCH_MARCH1(PBV, A, D) do { \
        if (has_s) hipLaunchKernelGGL((rhs25_march_kernel<TF, NJ, true, PBV, A, D>),  dim3(nblocks), dim3(64, NJ), 0, st, gd, mf, t); \
        else       hipLaunchKernelGGL((rhs25_march_kernel<TF, NJ, false, PBV, A, D>), dim3(nblocks), dim3(64, NJ), 0, st, gd, mf, t); } while (0)
    if (mode == 0)      { if (pb == 16) MHH_LAUNCH_MARCH(16); else if (pb == 4) MHH_LAUNCH_MARCH(4); else MHH_LAUNCH_MARCH(0); }
    else if (mode == 1) { if (pb == 16) MHH_LAUNCH_MARCH1(16, true, false); else MHH_LAUNCH_MARCH1(4, true, false); }      // one operator: LDS-DMA forms only
    else                { if (pb == 16) MHH_LAUNCH_MARCH1(16, false, true); else MHH_LAUNCH_MARCH1(4, false, true); }
#undef MHH_LAUNCH_MARCH1
#undef MHH_LAUNCH_MARCH
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
} // namespace

// entry used by mhh_rhs_exec for the (advec_2i5, diff_smag2) pair: u, v, w and scalar 0 (inputs validated by the caller)
int mhh_rhs25_march(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, void* stream)
{
    if (g->dtype == MHH_F64) return march_launch<double>(g, f, p, -1, -1, as_stream(stream));
    return march_launch<float>(g, f, p, -1, -1, as_stream(stream));
}
// the same over the rows [j0, j1) only (interior rows while the halos travel, edge rows after: mhh_rhs_exec_rows)
int mhh_rhs25_march_rows(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, void* stream)
{
    if (g->dtype == MHH_F64) return march_launch<double>(g, f, p, j0, j1, as_stream(stream));
    return march_launch<float>(g, f, p, j0, j1, as_stream(stream));
}

// Advec_2i5::exec / Diff_smag2::exec on their own, for u, v, w and scalar 0 (inputs validated by the caller): the marching
// kernel with one operator's terms only -- what the two calls of an unfused time step run.
int mhh_advec25_march(const mhh_grid* g, const mhh_fields* f, void* stream)
{
    if (g->dtype == MHH_F64) return march_launch<double>(g, f, nullptr, -1, -1, as_stream(stream), 1);
    return march_launch<float>(g, f, nullptr, -1, -1, as_stream(stream), 1);
}
int mhh_diff_smag2_march(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, void* stream)
{
    if (g->dtype == MHH_F64) return march_launch<double>(g, f, p, -1, -1, as_stream(stream), 2);
    return march_launch<float>(g, f, p, -1, -1, as_stream(stream), 2);
}

#ifdef MHH_MARCH_STAMPS
extern "C" __attribute__((visibility("default"))) int mhh_debug_march_stamps(unsigned long long* out)
{
    unsigned long long z[8] = {0,0,0,0,0,0,0,0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_march_stamps), sizeof(z)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_march_stamps), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}
#endif
