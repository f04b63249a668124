// k_march.hip -- the flagship kernel: advec_2i5 + diff_smag2 for u, v, w and one scalar in ONE pass, as a
// k-marching, LDS-staged, register-pipelined stencil sweep for gfx950.
//
// Why this shape (numbers: profiles/r1b_sweep_xcd_tiling.md): the one-thread-per-cell version issues ~180 vector
// loads per cell through the 64 B/clk L1 and is bound by that pipe, not by HBM or VALU. Here a 64 x NJ block of
// threads owns a column tile and walks up in k:
//   * the plane being updated lives in LDS with its halo (u, v, w, s: +-3; evisc: +-1, three levels), loaded once per
//     block by LDS-DMA (16-byte pieces on aligned rows, 4-byte pieces on any other layout) into a spare ring slot while
//     the current level is being computed: one barrier per level;
//   * each thread keeps the 6-level k-window (k-2 .. k+3) of its own column of u, v, w, s in registers -- the cell's own
//     values and its k-1 / k+1 neighbours come from there, not from LDS; u(i+1) and v(j+1) of level k-1, which the w
//     equation needs, are carried over from the previous level, so that the u, v rings hold the current plane only;
//   * vertical face fluxes (advective centred/upwind parts and the Smagorinsky stress) are computed once, on the
//     top face, and carried to the next level as its bottom face -- identical operands, identical rounding, so
//     the result is still bit-identical to the reference CPU path (src/advec_2i5.cxx, src/diff_smag2.cxx);
//   * interior levels run in groups of six with the register windows ROTATED at compile time (the level body is
//     instantiated for the six rotations): no register copies to shift four 6-level windows and twelve carried faces
//     per level (they were ~140 of the ~880 vector instructions per cell of the shifted form);
//   * the level body exists per (interior level or not, base-state density exactly 1 or not), so that on the hot path
//     the face orders, wall predicates and density factors are compile-time constants.
// Exact strength reductions (all checked bit for bit against the oracle, tests/test_parity.py):
//   * 0.5*(a+b) of an advecting velocity is folded into the metric (advec25_hor_f0, cell_ops.h);
//   * X * 2 * m is X * (2m); 0.5*s/tPr is s/(2 tPr);
//   * the division by the (uniform) turbulent Prandtl number is div_known (cell_ops.h): correctly rounded in 5 FMAs.
// Accumulation order per tendency is the reference's: t += advec_horizontal; t += advec_vertical; t += diffusion.
// Measurements and the experiments that did not pay: profiles/r1c_march_kernel_pmc.md, profiles/r1e_kernels_pmc.md,
// profiles/r2*_march*.md.
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include "k_common.h"
#include "k_march_common.h"
#include <gfx950_prims.h>   // angle form: the CPU emulation build (tests/emul) overrides it by include path

using namespace mhh;

#ifdef MHH_FMA_BUILD     // the named FMA build (build.py): its kernels carry their own name in profiler output
#define rhs25_march_kernel rhs25_march_fma_kernel
#endif

namespace
{
// 6-level register window of a column: logical index n holds level k-2+n; with rotation R it lives in w[(n+R) % 6]
template<int R, int N, class TF> __device__ __forceinline__ TF wv(const TF (&w)[6]) { return w[(N + R) % 6]; }
template<int R, class TF> __device__ __forceinline__ TF win_cen(const TF (&w)[6], int order)     // face k+1/2
{
    if (order == 2) return i2(wv<R,2>(w), wv<R,3>(w));
    if (order == 4) return i4ws(wv<R,1>(w), wv<R,2>(w), wv<R,3>(w), wv<R,4>(w));
    return i6(wv<R,0>(w), wv<R,1>(w), wv<R,2>(w), wv<R,3>(w), wv<R,4>(w), wv<R,5>(w));
}
template<int R, class TF> __device__ __forceinline__ TF win_upw(const TF (&w)[6], int order)
{
    if (order == 4) return i3ws(wv<R,1>(w), wv<R,2>(w), wv<R,3>(w), wv<R,4>(w));
    return i5(wv<R,0>(w), wv<R,1>(w), wv<R,2>(w), wv<R,3>(w), wv<R,4>(w), wv<R,5>(w));
}
// vertical advective increment from the face products T = rt*w_t*I_t, B = rb*w_b*I_b, Gt = rt*|w_t|*D_t, Gb likewise
// x / rc with the wave-uniform shortcut for rc == 1 (Boussinesq base state): x / 1 is x, bit for bit, and an
// fp64 division is ~12 VALU instructions that this kernel would otherwise issue ~12 times per cell.
template<class VT, class TF> __device__ __forceinline__ VT div_rho(VT x, TF rc, bool one) { return one ? x : x / rc; }

template<class VT, class TF> __device__ __forceinline__ VT vert_combine(int ot, int ob, VT T, VT B, VT Gt, VT Gb, TF rc, bool one, TF dz)
{
    VT cen;
    if (ob == 0)      cen = - div_rho( T, rc, one ) * dz;
    else if (ot == 0) cen = - div_rho( -B, rc, one ) * dz;
    else              cen = - div_rho( T - B, rc, one ) * dz;
    const bool ut = (ot >= 4), ub = (ob >= 4);
    if (ut && ub) return cen + div_rho( Gt - Gb, rc, one ) * dz;
    if (ut)       return cen + div_rho( Gt, rc, one ) * dz;
    if (ub)       return cen - div_rho( Gb, rc, one ) * dz;
    return cen;
}
template<class TF> __device__ __forceinline__ void shift6(TF (&w)[6], TF nw)
{
    w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = w[4]; w[4] = w[5]; w[5] = nw;
}

// The uniform coefficients of a level, as the kernel's FIRST argument: the kernel reads them from the kernel-argument segment with
// scalar loads inside every level (two groups of eight: what the u, v, w sections use / what the scalar's section uses) instead of
// holding all fourteen (28 scalar registers in fp64) across the whole kernel. With them live throughout, the register allocator
// spilled scalars to vector-register lanes and restored them with v_readlane -- a VECTOR instruction, the one issue resource this
// kernel is short of (36-65 per level and wave, PMC: kernel time follows the VALU count; profiles/r3_march_kernel.md).
template<class TF> struct alignas(16) MarchMetrics
{
    TF dxih, dyih;                               // 0.5 * TF(1.)/dx : advection of u, v, w (the 1/2 of the advecting-velocity mean folded in)
    TF dxd, dyd, dxd2, dyd2;                     // TF(1./dx) and twice that: diffusion
    TF visc, quarter;                            // quarter = 0.25 (see quarter_plus)
    TF dxi, dyi;                                 // TF(1.)/dx       : advection of the scalar
    TF dxidxi, dyidyi;                           // TF(1./(dx*dx)): diffusion of the scalar
    TF svisc;
    TF tPr2, rtPr2;                              // 2 tPr and RN(1 / (2 tPr)): 0.5*(a+b)/tPr is div_known(a+b, tPr2, rtPr2)
    TF pad1;
};

// 0.25*x + c in one instruction: the product with a power of two is exact (no underflow: eddy viscosities are zero or far above
// 2^-1020), so the fused multiply-add rounds once, exactly where the reference's addition rounds -- the same bits, one vector
// instruction less per four-point average of the eddy viscosity (ten per cell and level)
// Scheduling fences between the sections of a level (MHH_MARCH_FENCE=0 drops them): without them the scheduler hoists LDS reads
// across sections, shares eight operations between sections, and needs a register more than the 255 there are -- one scratch
// reload per level, i.e. a vmcnt(0) wait in the middle of the copies in flight: 5.06 against 4.74 ms (profiles/r3_march_kernel.md)
#ifndef MHH_MARCH_FENCE
#define MHH_MARCH_FENCE 1
#endif
__device__ __forceinline__ void mfence() { if constexpr (MHH_MARCH_FENCE != 0) sched_fence(); }
// (0.25 arrives as a scalar operand and c in a vector register: a VOP3 instruction takes one scalar register operand and no 32-bit
// literal on gfx9, and 0.25 is not an inline constant -- with both as scalars the compiler emitted v_mov + v_fmac, no saving)
template<class VT, class TF> __device__ __forceinline__ VT quarter_plus(VT x, TF quarter, VT c)
{
#ifndef MHH_MARCH_QP_F2
    if constexpr (lane_of<VT>::cells == 2) return TF(0.25)*x + c;        // two cells per lane: packed multiply + packed add (the fma form unpacks)
    else
#endif
    return tfma(x, VT(quarter), c);
}


template<class TF> struct MarchFields
{
    // hot: every level of the interior loop uses these
    const TF* __restrict__ u; const TF* __restrict__ v; const TF* __restrict__ w; const TF* __restrict__ s; const TF* __restrict__ ev;
    TF* __restrict__ ut; TF* __restrict__ vt; TF* __restrict__ wt; TF* __restrict__ st;
    // cold: walls, surface model, anelastic base state, folded buoyancy
    const TF* __restrict__ rhoref; const TF* __restrict__ rhorefh;
    const TF* __restrict__ ufb; const TF* __restrict__ uft; const TF* __restrict__ vfb; const TF* __restrict__ vft;
    const TF* __restrict__ sfb; const TF* __restrict__ sft;
    const TF* __restrict__ threfh; TF grav;      // folded dry buoyancy of the scalar (threfh == nullptr: off)
    int sm;
#ifdef MHH_MARCH_STAMP      // probe builds: per-wave cycle sums of the phases of a level (scripts/experiments/march_stamps.py)
    unsigned long long* dbg;
#endif
};
#ifdef MHH_MARCH_STAMP
#define MHH_STAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); stamp_acc[i] += t_ - stamp_last; stamp_last = t_; } while (0)
#else
#define MHH_STAMP(i) do {} while (0)
#endif

// x * 2^E for a wave-uniform, normal, non-zero x (a grid metric): an integer add on the exponent field, which stays on the
// scalar ALU -- gfx950 has no scalar fp64 multiply, and a vector one would park the uniform result in vector registers.
template<int E> __device__ __forceinline__ double scale2(double x)
{
    return __builtin_bit_cast(double, __builtin_bit_cast(long long, x) + (long long)E * (1LL << 52));
}
template<int E> __device__ __forceinline__ float scale2(float x)
{
    return __builtin_bit_cast(float, __builtin_bit_cast(int, x) + E * (1 << 23));
}

#ifndef MHH_MARCH_NJ
#define MHH_MARCH_NJ 4
#endif
#ifndef MHH_MARCH_OCC
#define MHH_MARCH_OCC 2           // blocks per CU the fp64 register budget is set for
#endif
#ifndef MHH_MARCH_OCC_F32
#define MHH_MARCH_OCC_F32 4       // fp32: half the registers and LDS -> four waves per SIMD (gabls1 1024x1024x256: 8.5 -> 7.9 ms)
#endif
#ifndef MHH_MARCH_UNROLL_F32
#define MHH_MARCH_UNROLL_F32 0    // fp32: 1 = the rotated six-level groups of the fp64 form
#endif
// PB = 16 : LDS-DMA in 16-byte pieces (global_load_lds_dwordx4; rows and fields 16-byte aligned, no staging registers, no
//           ds_write); PB = 4: in 4-byte pieces (global_load_lds_dword): any layout, four times the copy instructions.
// ADV / DIF: which operator's terms are added -- both (the fused pass), or one of them: Advec::exec and Diff::exec as
// separate calls then run the same kernel body (same bits, same order of accumulation as the fused pass in two steps).
// VT = the lane value: double or float (one cell per lane), or F2 (two fp32 cells per lane, packed arithmetic: cell_ops.h);
// TF = its scalar type (metrics, coefficients, the arrays in memory); CW = cells per lane, a wave spans 64*CW cells of a row.
// HX = cells the u, v, w, s tile starts west of the block's first cell: 3 (the stencil's reach) or 4 where that makes the tile's
// origin a whole 16-byte piece (e.g. 16 ghost cells in x: Grid::set_minimum_ghost_cells, src/grid.cxx:435-439)
template<class VT, int NJ, bool HAS_S, int PB, bool ADV = true, bool DIF = true, int HX = 3>
__global__ void __launch_bounds__(64*NJ, (sizeof(VT) == 4 ? MHH_MARCH_OCC_F32 : MHH_MARCH_OCC))
rhs25_march_kernel(const MarchMetrics<typename lane_of<VT>::scalar> mm, const GridDev<typename lane_of<VT>::scalar> g, const MarchFields<typename lane_of<VT>::scalar> f, const MarchTiling mt)
{
    using TF = typename lane_of<VT>::scalar;
    constexpr int CW = lane_of<VT>::cells;
    // CW = 2: a lane holds cells i and i + 64 of the wave's 128 (NOT neighbours: every LDS read of a plane is then two unit-stride
    // words across the lanes, free of bank conflicts -- neighbouring pairs read at odd offsets are stride-2 words, two-way conflicts
    // on 43 % of the LDS cycles of that form). SEC = distance to the lane's second cell; rows are whole tiles (imax % 128 == 0).
    constexpr int SEC = (CW == 2) ? 64 : 0;
    static_assert(PB == 16 || PB == 4, "piece size of the LDS-DMA copies");
    constexpr int VEC = 16 / (int)sizeof(TF);                       // elements per 16-byte DMA piece
    constexpr int AL = (PB == 16) ? VEC : 1;                        // granularity of tile widths / origins in elements
    constexpr int TI = ((64*CW + HX + 3 + AL-1)/AL)*AL;             // u,v,w,s tile: x from i0-HX
    constexpr int EX = (PB == 16) ? VEC : 1;                        // evisc tile: x from i0-EX (aligned for 16-byte DMA)
    constexpr int TE = ((64*CW + EX + 1 + AL-1)/AL)*AL;
    constexpr int TJ = NJ + 6, TJE = NJ + 2, NT = 64*NJ;
    // A tile slot in LDS is a whole number of wave sweeps (64 pieces of PB bytes): the last sweep of a tile copies (clamped, valid)
    // duplicates into the slot's padding instead of running under a lane mask -- per tile and level that mask was two v_readlane of
    // a spilled exec mask, four scalar instructions and a branch
#ifndef MHH_MARCH_PAD
#define MHH_MARCH_PAD 1
#endif
    constexpr bool PAD = (MHH_MARCH_PAD != 0);
    constexpr int NTILE = PAD ? ((TI*TJ*(int)sizeof(TF) + 64*PB - 1) / (64*PB)) * (64*PB) / (int)sizeof(TF) : TI*TJ;
    constexpr int NETILE = PAD ? ((TE*TJE*(int)sizeof(TF) + 64*PB - 1) / (64*PB)) * (64*PB) / (int)sizeof(TF) : TE*TJE;
    // LDS rings, one slot deeper than what a level reads so that the copy of the next plane runs under the whole compute
    // phase: u, v, s: level k; w: k and k+1; evisc: k-1..k+1. One array, so that every plane is a compile-time offset from
    // one base (ds_read immediates; the LDS-DMA destination is base + constant).
    constexpr int RU = 2, RW = 3, RE = DIF ? 4 : 0, RS = HAS_S ? 2 : 0;
    constexpr int OU = 0, OV = OU + RU*NTILE, OW = OV + RU*NTILE, OS = OW + RW*NTILE, OE = OS + RS*NTILE, LTOT = OE + RE*NETILE;
    __shared__ __attribute__((aligned(16))) TF L[LTOT];

    int bx, by, kcn;
    if (!decode_march(mt, blockIdx.x, bx, by, kcn)) return;        // whole block leaves together: no barrier hazard
    const int jj = g.icells, kk = g.ijcells;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty*64 + tx;
    int j0, jlim; march_tile_rows(mt, by, NJ, j0, jlim);
    const int i0 = g.istart + bx*64*CW;
    const int kb = g.kstart + kcn*mt.kc;
    const int ke = (kb + mt.kc < g.kend) ? kb + mt.kc : g.kend;
    const int i = i0 + tx, j = j0 + ty;                             // the lane's (first) cell; CW = 2 needs whole 128-cell tiles (the launcher checks)
    const bool active = (i + SEC < g.iend) && (j < jlim);
    const int ci = (i + SEC < g.iend) ? i : g.iend-1-SEC, cj = (j < jlim) ? j : jlim-1;   // clamped column for the window loads
    const int col = ci + cj*jj;
    const int ij = col;
    const int l = (ty+3)*TI + (tx+HX), le = (ty+1)*TE + (tx+EX);
    // a plane in LDS as seen from the lane's cell: [o] = the lane value o cells away (CW = 2: the lane's two cells, one ds_read2_b32)
    struct LV
    {
        const TF* p;
        __device__ __forceinline__ VT operator[](int o) const { if constexpr (CW == 1) return p[o]; else return VT(p[o], p[o+SEC]); }
    };
    auto ld2d = [&](const TF* q, int c) -> VT { if constexpr (CW == 1) return q[c]; else return VT(q[c], q[c+SEC]); };   // a 2-D surface array at the lane's column

    // interior level of an updating iteration: faces k and k+1 of the centred fields and the w "faces" k-1, k all 6th order
    int kf0 = (kb > g.kstart+3) ? kb : g.kstart+3;                 // first interior level of the chunk
    int kf1 = (ke < g.kend-3) ? ke : g.kend-3;                     // one past its last
#ifdef MHH_MARCH_NO_FAST    // A/B builds only
    kf0 = kf1 = ke;
#endif
    if (kf1 < kf0) kf1 = kf0;
    if (kf0 > ke) kf0 = kf1 = ke;
    // ring slot of plane p: counted from the first level of the rotated groups, so that inside a group of six the slots of
    // the 2- and 3-deep rings are compile-time constants (12 is a multiple of every ring depth; planes from kb-2 on are
    // copied: p - kg0 >= -5)
    const int kg0 = (kf0 - kb <= 3) ? kf0 : kb;
    auto slot = [&](int p, int r) { const unsigned q = (unsigned)(p - kg0 + 12); return (int)((r & (r-1)) == 0 ? (q & (unsigned)(r-1)) : q % (unsigned)r); };

    // ---- wave-uniform values of the interior loop, each pinned in scalar registers of its own. Without this the compiler
    // keeps the kernel-argument structs as the 16-dword tuples its merged s_loads produced, spills them whole and restores
    // all sixteen dwords (v_readlane: a VECTOR instruction each) to use one pointer -- ~190 of them per level. ------------
    const size_t kk8 = sgpr((size_t)kk * sizeof(TF));              // bytes per plane
    auto adv = [&](const TF* q, int n) -> const TF* { return reinterpret_cast<const TF*>(reinterpret_cast<const char*>(q) + n*kk8); };
    // the uniform coefficients: read per level from the kernel-argument segment (see MarchMetrics)
    const TF* const kmm = first_kernarg(reinterpret_cast<const TF&>(mm));
    const TF* __restrict__ tdzi = sgpr(g.dzi); const TF* __restrict__ tdzhi = sgpr(g.dzhi);

    // ---- tile movers. A tile is walked in pieces of PW 32-bit words: e = tid + n*NT; piece -> (row, first word) --------
    constexpr int EW = (int)sizeof(TF) / 4;                           // words per element
    constexpr int PW = PB/4;                                          // words per piece
    constexpr int PPR = TI*EW / PW, PPRE = TE*EW / PW;                // pieces per tile row
    constexpr int NP = PPR*TJ, NPE = PPRE*TJE;
    constexpr int NLD = (NP + NT - 1) / NT, NLDE = (NPE + NT - 1) / NT;
    // Every lane's piece lies inside the array: tiles that stick out over the east / north edge of the array (ragged grids)
    // copy the array's last pieces / rows instead -- those LDS cells are read by inactive threads only -- so no lane mask is
    // needed but for the one wave per sweep that straddles the end of the tile.
    unsigned off[NLD], offe[NLDE > 0 ? NLDE : 1];                     // byte offsets from the start of a plane
#pragma unroll
    for (int n=0; n<NLD; ++n)
    {
        const int e = (tid + n*NT < NP) ? tid + n*NT : NP-1;
        const int tj = e / PPR, tw = (e - tj*PPR)*PW;
        int gw = (i0 - HX)*EW + tw, gj = j0 - 3 + tj;
        if (gw + PW > g.icells*EW) gw = g.icells*EW - PW;
        if (gj >= g.jcells) gj = g.jcells - 1;
        off[n] = (unsigned)(gw + gj*jj*EW) * 4u;
    }
#pragma unroll
    for (int n=0; n<NLDE; ++n)
    {
        const int e = (tid + n*NT < NPE) ? tid + n*NT : NPE-1;
        const int tj = e / PPRE, tw = (e - tj*PPRE)*PW;
        int gw = (i0 - EX)*EW + tw, gj = j0 - 1 + tj;
        if (gw + PW > g.icells*EW) gw = g.icells*EW - PW;
        if (gj >= g.jcells) gj = g.jcells - 1;
        offe[n] = (unsigned)(gw + gj*jj*EW) * 4u;
    }
    // LDS-DMA movers: PB bytes per lane straight into the ring slot
    const int wave_e0 = tid & ~63;                                    // first piece index of this wave within a sweep
    // SV: the copy as scalar base + per-lane byte offset + scalar LDS address (gfx950_prims.h): no vector ALU per piece
#ifdef MHH_DMA_NO_SV
    constexpr bool SV = false;
#else
    constexpr bool SV = (MHH_RAW_DMA != 0);
#endif
    const unsigned lds_wave = sgpr(lds_address(L) + uniform_u32((unsigned)(wave_e0*PW*4)));   // this wave's lane-0 byte address in L
    // lo = element offset of the destination tile in L
    auto dma_piece = [&](const TF* plane, unsigned byte_off, int lo, int n)
    {
        if constexpr (SV) lds_dma_sv2<PB>(plane, byte_off, lds_wave, (unsigned)(lo*(int)sizeof(TF) + n*NT*PW*4));
        else
        {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(plane) + byte_off/4u;
            uint32_t* dst = reinterpret_cast<uint32_t*>(L + lo) + (size_t)(wave_e0 + n*NT)*PW;
            if constexpr (PB == 16) lds_dma16<(sizeof(TF) == 8)>(src, dst); else lds_dma4<(sizeof(TF) == 8)>(src, dst);
        }
    };
    // pl = the plane to copy (a wave-uniform pointer inside the array)
    // a wave takes part in sweep n if its first piece lies inside the tile: a wave-uniform (scalar) condition
    const int wave_p0 = (int)uniform_u32((unsigned)wave_e0);
    auto dma_tile = [&](const TF* __restrict__ pl, int lo)
    {
#pragma unroll
        for (int n=0; n<NLD; ++n)
            if ((n+1)*NT <= NP || (PAD ? wave_p0 + n*NT < NP : tid + n*NT < NP)) dma_piece(pl, off[n], lo, n);
    };
    auto dma_etile = [&](const TF* __restrict__ pl, int lo)
    {
#pragma unroll
        for (int n=0; n<NLDE; ++n)
            if ((n+1)*NT <= NPE || (PAD ? wave_p0 + n*NT < NPE : tid + n*NT < NPE)) dma_piece(pl, offe[n], lo, n);
    };
    const int kmaxp = g.kcells - 1;                                   // planes are clamped into [0, kcells-1]: values read from a
    auto plane = [&](const TF* __restrict__ fld, int kp) -> const TF*   // clamped plane are never used by an updated level
    {
        const int kq = kp < 0 ? 0 : (kp > kmaxp ? kmaxp : kp);
        return fld + (size_t)kq*kk;
    };
    // column values and tendencies go through scalar base + 32-bit lane byte offset (gload / gstore, gfx950_prims.h).
    // Tendencies are touched once per kernel: their loads and stores carry the non-temporal hint, so that they stream past L2
    // instead of evicting the field planes that neighbouring tiles re-read (512^3: HBM fetch 15.4 -> 14.1 GB per launch, same
    // time; -DMHH_MARCH_NO_NT for A/B runs)
    auto gl = [](const TF* base, unsigned bo) -> VT { if constexpr (CW == 1) return gload(base, bo); else return VT(gload(base, bo), gload(base, bo + 4u*SEC)); };
#if defined(MHH_EXP_NOTEND) || defined(MHH_EXP_NOMEM)   // diagnostic builds (fp64): no tendency traffic / the arithmetic alone (stores behind a condition that never holds)
    auto tld = [](const TF*, unsigned bo) -> VT { return VT(TF(bo)); };
    auto tst = [](TF* base, unsigned bo, VT v) { if constexpr (CW == 1) { if (v == TF(-1.2345e300)) gstore(base, bo, v); } };
#elif !defined(MHH_MARCH_NO_NT)
    auto tld = [](const TF* base, unsigned bo) -> VT { if constexpr (CW == 1) return gload_stream(base, bo); else return VT(gload_stream(base, bo), gload_stream(base, bo + 4u*SEC)); };
    auto tst = [](TF* base, unsigned bo, VT v) { if constexpr (CW == 1) gstore_stream(base, bo, v); else { gstore_stream(base, bo, v.lo()); gstore_stream(base, bo + 4u*SEC, v.hi()); } };
#else
    auto tld = [](const TF* base, unsigned bo) -> VT { if constexpr (CW == 1) return gload(base, bo); else return VT(gload(base, bo), gload(base, bo + 4u*SEC)); };
    auto tst = [](TF* base, unsigned bo, VT v) { if constexpr (CW == 1) gstore(base, bo, v); else { gstore(base, bo, v.lo()); gstore(base, bo + 4u*SEC, v.hi()); } };
#endif
    unsigned bo0 = (unsigned)col * (unsigned)sizeof(TF);              // this column in the plane of the running level (advances by a plane per level)

    // ---- prologue: the planes level ks reads, windows centred on ks ----------------------------------------------------
    const int ks = kb - 1;                 // warm-up level: only top-face quantities are formed there
    dma_tile(plane(f.u, ks), OU + slot(ks, RU)*NTILE); dma_tile(plane(f.v, ks), OV + slot(ks, RU)*NTILE);
    dma_tile(plane(f.w, ks), OW + slot(ks, RW)*NTILE); dma_tile(plane(f.w, ks+1), OW + slot(ks+1, RW)*NTILE);
    if constexpr (DIF) for (int p = ks-1; p <= ks+1; ++p) dma_etile(plane(f.ev, p), OE + slot(p, RE)*NETILE);
    if constexpr (HAS_S) dma_tile(plane(f.s, ks), OS + slot(ks, RS)*NTILE);
    VT uw[6], vw[6], ww[6], sw[6];         // levels ks-2 .. ks+3
#pragma unroll
    for (int n=0; n<6; ++n)
    {
        uw[n] = ld2d(plane(f.u, ks-2+n), col); vw[n] = ld2d(plane(f.v, ks-2+n), col); ww[n] = ld2d(plane(f.w, ks-2+n), col);
        sw[n] = HAS_S ? ld2d(plane(f.s, ks-2+n), col) : VT(TF(0));
    }
    // running plane pointers (wave-uniform): at level k, pu / pv / ps point at plane k+1, pw / pe at plane k+2 (the planes
    // to copy; the window values of level k+4 are three / two planes further up), the tendencies' at plane k
    // Plane BASES of the chunk, constant through the kernel; the level enters through the lanes' byte offsets, which advance by one
    // plane per level (seven 32-bit vector adds). Running scalar POINTERS were rewritten by the loop optimiser into spilled bases
    // plus m x stride and cost 15 v_readlane and ~40 scalar instructions per level. (32-bit offsets: the host keeps
    // (kc + 8) planes below 4 GB, march_launch.)
    const TF* const pu = sgpr(f.u + (size_t)(ks+1)*kk); const TF* const pv = sgpr(f.v + (size_t)(ks+1)*kk); const TF* const pw = sgpr(f.w + (size_t)(ks+2)*kk);
    const TF* const ps = HAS_S ? sgpr(f.s + (size_t)(ks+1)*kk) : nullptr; const TF* const pe = DIF ? sgpr(f.ev + (size_t)(ks+2)*kk) : nullptr;
    TF* const put = sgpr(f.ut + (long long)ks*kk); TF* const pvt = sgpr(f.vt + (long long)ks*kk); TF* const pwt = sgpr(f.wt + (long long)ks*kk);
    TF* const pst = HAS_S ? sgpr(f.st + (long long)ks*kk) : nullptr;
    const unsigned kk8w = sgpr((unsigned)kk8);                     // bytes per plane as a 32-bit scalar
    wait_vmem();
    __syncthreads();

    // carried bottom-face products: advective centred (T) and upwind (G) parts, diffusive flux (D); the advective ones of
    // u, v, w hold TWICE the reference's face product (the advecting velocity enters as the sum of the two values it averages)
    const VT zero = VT(TF(0));
    VT cTu = zero, cGu = zero, cDu = zero, cTv = zero, cGv = zero, cDv = zero, cTw = zero, cGw = zero, cDw = zero, cTs = zero, cGs = zero, cDs = zero;
    VT u1m = zero, vNm = zero;             // u(i+1), v(j+1) of the level below (the w equation's faces)

    // Latency of the tendency read-modify-writes (fp64 form; the fp32 form has no registers to spare at four waves per SIMD):
    //  * TPREF: the tendencies of the NEXT level are loaded a whole level ahead of their use, like the LDS-DMA planes;
    //  * DSTORE: the last tendency finished in a level (the scalar's) is stored at the top of the next level, so that the
    //    s_waitcnt vmcnt(0) in front of the barrier -- which on gfx9 also waits for stores -- does not find it just issued.
#ifndef MHH_MARCH_TPREF
#define MHH_MARCH_TPREF 1
#endif
#ifndef MHH_MARCH_DSTORE
#define MHH_MARCH_DSTORE 1
#endif
    constexpr bool TPREF = (sizeof(VT) == 8) && (MHH_MARCH_TPREF != 0);
    constexpr bool DSTORE = (sizeof(VT) == 8) && HAS_S && (MHH_MARCH_DSTORE != 0);
    VT dss = zero; bool dsp = false;       // deferred scalar tendency of the level below
    VT tpu = zero, tpv = zero, tpw = zero, tps = zero;

#ifdef MHH_MARCH_STAMP
    unsigned long long stamp_acc[5] = {0, 0, 0, 0, 0}, stamp_last = __builtin_readcyclecounter();
#endif
    // One level. FAST = an interior level of an updating iteration: every vertical face is 6th/5th order, no wall or
    // surface-flux branch applies -- the face orders and the wall predicates become constants and their dispatch
    // (a third of the loop's scalar / control instructions) disappears. Same arithmetic either way.
    // RHO1 = rhoref and rhorefh are exactly 1 on every level this block touches (Boussinesq base state): 1*x and x/1 are x,
    // bit for bit, so the density factors and divisions drop out of the instantiation instead of being tested per use.
    // ROT = rotation of the register windows (0..5), or -1: rotation 0 and a physical shift at the end of the level.
    auto level = [&](const int k, auto fast_tag, auto rho1_tag, auto rot_tag) __attribute__((always_inline))
    {
        constexpr bool FAST = decltype(fast_tag)::value, RHO1 = decltype(rho1_tag)::value;
        constexpr int ROT = decltype(rot_tag)::value, RR = (ROT < 0) ? 0 : ROT;
        auto R = [](TF r, VT x) -> VT { return RHO1 ? x : r*x; };
        // ring slot of plane k+d: inside a rotated group k - kg0 = ROT (mod 6), a constant for the rings whose depth divides 6
        auto sl = [&](int d, int r) { return (ROT >= 0 && 6 % r == 0) ? (ROT + d + 12) % r : slot(k + d, r); };
        const TF* const mq = sgpr(kmm);                          // opaque per level: the loads stay inside the level
        const Uniform8<TF> mg0 = uniform_load8(mq);
        const TF dxih = mg0.v[0], dyih = mg0.v[1], dxd = mg0.v[2], dyd = mg0.v[3], dxd2 = mg0.v[4], dyd2 = mg0.v[5], visc = mg0.v[6], quarter = mg0.v[7];
        VT viscv = VT(visc); if constexpr (CW == 1) pin_vgpr(viscv); else pin_vgpr(viscv.v);                    // the viscosity once per level in a vector register (quarter_plus)
        MHH_STAMP(4);                                             // (loop control, window rotation: since the last barrier)
        // ---- moving the next level's planes: k+1 of u, v, s; k+2 of w, evisc; window value k+4 ---------------------------
        // The four waves of a block leave the barrier together, and a wave issues in order: with all copies at the top of the
        // level every wave sat out the block's whole burst of vector-memory instructions in the address unit's queue before its
        // first arithmetic (cycle stamps, profiles/r3_march_kernel.md: ~1500 of a level's ~8500 wave cycles). The copies are
        // therefore issued in GROUPS between the sections of the level (MHH_MARCH_SPREAD: 0 = all at the top, as in round 2).
        // (the test `k + 1 < ke` is spelled out where it is used: as a bool carried between basic blocks it became a lane mask rebuilt
        //  with two vector instructions per use)
        // k + 1 < ke: k+2 <= ke <= kend lies inside the array
        auto copy_group = [&](int grp) __attribute__((always_inline))
        {
#ifndef MHH_EXP_NOMEM
            if (k + 1 < ke)
            {
                if (grp == 0) { dma_tile(pu, OU + sl(1, RU)*NTILE); dma_tile(pv, OV + sl(1, RU)*NTILE); }
                if (grp == 1) { dma_tile(pw, OW + sl(2, RW)*NTILE); if constexpr (DIF) dma_etile(pe, OE + sl(2, RE)*NETILE); }
                if (grp == 2) { if constexpr (HAS_S) dma_tile(ps, OS + sl(1, RS)*NTILE); }
            }
#endif
        };
        // where group g is issued: position 0 = top of the level, 1 = after the advective top faces, 2 = after the diffusive top
        // faces (all ahead of the masked update of the tendencies: every lane issues its pieces)
#ifndef MHH_MARCH_SPREAD
#define MHH_MARCH_SPREAD 1
#endif
        // (fp64 only: the two-cells-per-lane fp32 form measured 5.53 ms spread against 5.18 ms with all copies at the top, gabls1 1024 x 1024 x 256)
        constexpr int SPREAD = (sizeof(TF) == 8) ? MHH_MARCH_SPREAD : 0;
        auto copies_at = [&](int pos) __attribute__((always_inline))
        {
            constexpr int where[3][3] = { {0, 0, 0}, {0, 1, 2}, {1, 2, 2} };
            bool any = false;
            for (int g_ = 0; g_ < 3; ++g_) if (where[SPREAD][g_] == pos) any = true;
            if (!any) return;
            if (pos > 0) mfence();
            for (int g_ = 0; g_ < 3; ++g_) if (where[SPREAD][g_] == pos) copy_group(g_);
            mfence();
        };
        copies_at(0);
        // window values of level k+4, read unconditionally (no select, no copy): past the top of the array the plane pointer
        // steps back onto the last plane -- such values only enter faces above the top wall, which are never formed
        // (an interior level has k + 4 <= kend <= kcells - 1: nothing to step back, and no 64-bit scalar multiply per level)
        const int over = FAST ? 0 : ((k + 4 > kmaxp) ? kmaxp - (k + 4) : 0);    // <= 0
#if !defined(MHH_EXP_NOMEM) && !defined(MHH_EXP_NOCOL)
        // one / two / three planes up: derived where they are used (three adds, as many as keeping them as running values costs,
        // and three vector registers less across the level)
        const unsigned bo1 = bo0 + kk8w, bo2 = bo0 + 2u*kk8w, bo3 = bo0 + 3u*kk8w;
        const unsigned ovb = FAST ? 0u : (unsigned)over * kk8w;    // (two's complement: a step back of -over planes)
        const VT nu = gl(pu, bo3 + ovb), nv = gl(pv, bo3 + ovb), nw = gl(pw, bo2 + ovb);
        const VT ns = HAS_S ? gl(ps, bo3 + ovb) : zero;
#else
        const VT nu = VT(TF(bo3 + over) * TF(1e-3)), nv = nu + TF(1), nw = nu - TF(1), ns = nu + TF(2);
#endif
        if (DSTORE && dsp && active) tst(pst, bo0 - kk8w, dss);   // the scalar tendency of level k-1
        dsp = false;
        const VT tcu = tpu, tcv = tpv, tcw = tpw, tcs = tps;      // this level's tendencies (loaded during the previous level)
        if constexpr (TPREF) {             // the warm-up level ks = kb-1 fetches those of kb. Unconditional: inactive lanes sit on
            // a clamped (valid) column and plane k+1 <= kend exists, so no lane mask, no select, no register copy
            tpu = tld(put, bo1); tpv = tld(pvt, bo1); tpw = tld(pwt, bo1); if constexpr (HAS_S) tps = tld(pst, bo1); }
        MHH_STAMP(0);                                             // copies and loads of the next level issued

        const LV uk{L + OU + sl(0, RU)*NTILE + l};
        const LV vk{L + OV + sl(0, RU)*NTILE + l};
        const LV wk{L + OW + sl(0, RW)*NTILE + l}, wkp{L + OW + sl(1, RW)*NTILE + l};
        const LV sk{L + (HAS_S ? OS + sl(0, RS ? RS : 1)*NTILE + l : 0)};
        const LV ek{L + (DIF ? OE + sl(0, RE ? RE : 1)*NETILE + le : 0)};
        const LV ekm{L + (DIF ? OE + sl(-1, RE ? RE : 1)*NETILE + le : 0)};
        const LV ekp{L + (DIF ? OE + sl(1, RE ? RE : 1)*NETILE + le : 0)};
        // the own column at k-1, k, k+1: from the register windows
        const VT u0m = wv<RR,1>(uw), u0 = wv<RR,2>(uw), u0p = wv<RR,3>(uw);
        const VT v0m = wv<RR,1>(vw), v0 = wv<RR,2>(vw), v0p = wv<RR,3>(vw);
        const VT w0 = wv<RR,2>(ww), w0p = wv<RR,3>(ww);
        const VT s0m = wv<RR,1>(sw), s0 = wv<RR,2>(sw), s0p = wv<RR,3>(sw);
        // per-level coefficients: scalar loads (uniform_load), not vector loads whose wait would drain the copies in flight
        const TF rhkp = RHO1 ? TF(1) : uniform_load(f.rhorefh, k+1), rhk = RHO1 ? TF(1) : uniform_load(f.rhorefh, k), rk = RHO1 ? TF(1) : uniform_load(f.rhoref, k);
        const TF dzi = uniform_load(tdzi, k), dzhi = uniform_load(tdzhi, k), dzhip = uniform_load(tdzhi, k+1);
        const TF dzih = scale2<-1>(dzi), dzhih = scale2<-1>(dzhi), dzhi2 = scale2<1>(dzhi);
        const bool rk1 = RHO1 || (rk == TF(1.)), rhk1 = RHO1 || (rhk == TF(1.));
        const int otc = FAST ? 6 : order_face_c(k+1, g.kstart, g.kend);
        const int obc = FAST ? 6 : order_face_c(k, g.kstart, g.kend);
        const bool wlev = FAST || (k >= g.kstart);               // the w equation's "faces" are cell centres kstart..kend-1
        const int otw = FAST ? 6 : (wlev ? order_face_w(k, g.kstart, g.kend) : 0);
        const int obw = FAST ? 6 : ((k-1 >= g.kstart) ? order_face_w(k-1, g.kstart, g.kend) : 0);
        // surface model: the lowest / highest level takes the prescribed flux instead of the resolved one
        const bool fb = !FAST && f.sm && (k == g.kstart), ft = !FAST && f.sm && (k == g.kend-1);
        const bool need_dtop = FAST || (!(ft) && (k < g.kend-1 || !f.sm) && (k+1 <= g.kend));   // top diffusive flux of level k is used by k or k+1

        const VT u_e = uk[1], v_n = vk[TI];                      // also next level's u1m, vNm

        // ---- top-face quantities of level k --------------------------------------------------------------------
        VT Tu = zero, Gu = zero, Tv = zero, Gv = zero, Tw = zero, Gw = zero, Ts = zero, Gs = zero;
        if (ADV && otc != 0)
        {
            const VT swu = wkp[-1] + w0p;                        // 2 x the advecting w at the u / v point
            const VT swv = wkp[-TI] + w0p;
            Tu = R(rhkp, swu) * win_cen<RR>(uw, otc);
            Tv = R(rhkp, swv) * win_cen<RR>(vw, otc);
            if (otc >= 4) { Gu = R(rhkp, tabs(swu)) * win_upw<RR>(uw, otc); Gv = R(rhkp, tabs(swv)) * win_upw<RR>(vw, otc); }
            if (HAS_S)
            {
                Ts = R(rhkp, w0p) * win_cen<RR>(sw, otc);
                if (otc >= 4) Gs = R(rhkp, tabs(w0p)) * win_upw<RR>(sw, otc);
            }
        }
        if (ADV && wlev)
        {
            const VT sww = w0 + w0p;
            Tw = R(rk, sww) * win_cen<RR>(ww, otw);
            if (otw >= 4) Gw = R(rk, tabs(sww)) * win_upw<RR>(ww, otw);
        }
        copies_at(1);
        VT Du = zero, Dv = zero, Dw = zero, Ds = zero;
        if (DIF && need_dtop)
        {
            const VT etu = quarter_plus(ek[-1] + ek[0] + ekp[-1] + ekp[0], quarter, viscv);
            Du = R(rhkp, etu)*((u0p-u0)*dzhip + (w0p-wkp[-1])*dxd);
            const VT etv = quarter_plus(ek[-TE] + ek[0] + ekp[-TE] + ekp[0], quarter, viscv);
            Dv = R(rhkp, etv)*((v0p-v0)*dzhip + (w0p-wkp[-TI])*dyd);
        }
        if (DIF && wlev)
        {
            const VT etw = ek[0] + visc;
            Dw = R(rk, etw)*(w0p-w0)*dzi;
        }

        copies_at(2);
#ifdef MHH_EXP_NOMATH          // diagnostic build: the memory traffic alone (copies, loads, stores; one LDS read per plane)
        if ((FAST || k >= kb) && active)
        {
            tst(put, bo0, (TPREF ? tcu : tld(put, bo0)) + uk[0] + ek[0]); tst(pvt, bo0, (TPREF ? tcv : tld(pvt, bo0)) + vk[0]);
            tst(pwt, bo0, (TPREF ? tcw : tld(pwt, bo0)) + wk[0] + wkp[0]); if constexpr (HAS_S) tst(pst, bo0, (TPREF ? tcs : tld(pst, bo0)) + sk[0]);
        }
        const bool upd = false;
#else
        const bool upd = (FAST || k >= kb) && active;
#endif
        // ---- update the tendencies of level k ----------------------------------------------------------------
        // (sched_fence between the field sections: the instruction scheduler otherwise hoists every LDS read of a level to
        //  its top and the level needs more than the 256 registers of two waves per SIMD; a section is a block of its own under the
        //  lane mask, so that the copy groups between sections are issued by every lane)
        // One section per tendency, each a block of its own under the lane mask. `which`: 0 = u, 1 = v, 2 = w, 3 = the scalar.
        auto section = [&](const int which) __attribute__((always_inline))
        {
            if (which == 0)
            {
                if (upd)
                {   // u
                    VT t = TPREF ? tcu : tld(put, bo0);
                    if constexpr (ADV)
                    {
                        t += advec25_hor_f0(uk, u0, TI, u0 + u_e, uk[-1] + u0, vk[TI-1] + v_n, vk[-1] + v0, dxih, dyih);
                        t += vert_combine(otc, obc, Tu, cTu, Gu, cGu, rk, rk1, dzih);
                    }
                    if constexpr (DIF)
                    {
                        const VT ee = ek[0] + visc, ew = ek[-1] + visc;
                        const VT en = quarter_plus(ek[-1   ] + ek[0  ] + ek[-1+TE] + ek[TE], quarter, viscv);
                        const VT es = quarter_plus(ek[-1-TE] + ek[-TE] + ek[-1   ] + ek[0 ], quarter, viscv);
                        const VT hor = + ( ee*(u_e-u0)*dxd - ew*(u0-uk[-1])*dxd ) * dxd2
                                       + ( en*((uk[TI]-u0    )*dyd + (v_n-vk[TI-1])*dxd)
                                         - es*((u0    -uk[-TI])*dyd + (v0 -vk[-1  ])*dxd) ) * dyd;
                        VT ver;
                        if (fb)      ver = div_rho( Du + rhk * ld2d(f.ufb, ij), rk, rk1 ) * dzi;
                        else if (ft) ver = div_rho( - rhkp * ld2d(f.uft, ij) - cDu, rk, rk1 ) * dzi;
                        else         ver = div_rho( Du - cDu, rk, rk1 ) * dzi;
                        t += hor + ver;
                    }
                    tst(put, bo0, t);
                }
            }
            else if (which == 1)
            {
                if (upd)
                {   // v
                    VT t = TPREF ? tcv : tld(pvt, bo0);
                    if constexpr (ADV)
                    {
                        t += advec25_hor_f0(vk, v0, TI, uk[1-TI] + u_e, uk[-TI] + u0, v0 + v_n, vk[-TI] + v0, dxih, dyih);
                        t += vert_combine(otc, obc, Tv, cTv, Gv, cGv, rk, rk1, dzih);
                    }
                    if constexpr (DIF)
                    {
                        const VT ee = quarter_plus(ek[-TE  ] + ek[0 ] + ek[1-TE] + ek[1], quarter, viscv);
                        const VT ew = quarter_plus(ek[-1-TE] + ek[-1] + ek[-TE ] + ek[0], quarter, viscv);
                        const VT en = ek[0] + visc, es = ek[-TE] + visc;
                        const VT hor = + ( ee*((vk[1]-v0    )*dxd + (u_e-uk[1-TI])*dyd)
                                         - ew*((v0   -vk[-1])*dxd + (u0 -uk[-TI ])*dyd) ) * dxd
                                       + ( en*(v_n-v0)*dyd - es*(v0-vk[-TI])*dyd ) * dyd2;
                        VT ver;
                        if (fb)      ver = div_rho( Dv + rhk * ld2d(f.vfb, ij), rk, rk1 ) * dzi;
                        else if (ft) ver = div_rho( - rhkp * ld2d(f.vft, ij) - cDv, rk, rk1 ) * dzi;
                        else         ver = div_rho( Dv - cDv, rk, rk1 ) * dzi;
                        t += hor + ver;
                    }
                    tst(pvt, bo0, t);
                }
            }
            else if (which == 2)
            {
                if (upd)
                {
                    if (FAST || k > g.kstart)
                    {   // w
                        VT t = TPREF ? tcw : tld(pwt, bo0);
                        if (HAS_S && f.threfh) { const TF th_k = uniform_load(f.threfh, k); t += f.grav/th_k * (i2(s0m, s0) - th_k); }   // src/thermo_dry.cxx:165-178
                        if constexpr (ADV)
                        {
                            t += advec25_hor_f0(wk, w0, TI, u1m + u_e, u0m + u0, vNm + v_n, v0m + v0, dxih, dyih);
                            t += vert_combine(otw, obw, Tw, cTw, Gw, cGw, rhk, rhk1, dzhih);
                        }
                        if constexpr (DIF)
                        {
                            const VT ee = quarter_plus(ekm[0  ] + ek[0  ] + ekm[1 ] + ek[1 ], quarter, viscv);
                            const VT ew = quarter_plus(ekm[-1 ] + ek[-1 ] + ekm[0 ] + ek[0 ], quarter, viscv);
                            const VT en = quarter_plus(ekm[0  ] + ek[0  ] + ekm[TE] + ek[TE], quarter, viscv);
                            const VT es = quarter_plus(ekm[-TE] + ek[-TE] + ekm[0 ] + ek[0 ], quarter, viscv);
                            t += + ( ee*((wk[1 ]-w0     )*dxd + (u_e-u1m)*dzhi)
                                   - ew*((w0    -wk[-1 ])*dxd + (u0 -u0m)*dzhi) ) * dxd
                                 + ( en*((wk[TI]-w0     )*dyd + (v_n-vNm)*dzhi)
                                   - es*((w0    -wk[-TI])*dyd + (v0 -v0m)*dzhi) ) * dyd
                                 + div_rho( Dw - cDw, rhk, rhk1 ) * dzhi2;
                        }
                        tst(pwt, bo0, t);
                    }
                }
            }
            else
            {
                // the scalar: its coefficients arrive with the section's LDS reads; the diffusive flux through the top face is formed on
                // every level (the next level carries it), the tendency on updating levels
                if constexpr (HAS_S)
                {
            const Uniform8<TF> mg1 = uniform_load8(mq + 8);
                    const TF dxi = mg1.v[0], dyi = mg1.v[1], dxidxi = mg1.v[2], dyidyi = mg1.v[3], svisc = mg1.v[4], tPr2 = mg1.v[5], rtPr2 = mg1.v[6];
                    auto div_tpr = [&](VT x) -> VT { return div_known(x, tPr2, rtPr2); };     // 0.5*x / tPr
                    if (DIF && need_dtop)
                    {
                        const VT ets = div_tpr(ek[0]+ekp[0]) + svisc;
                        Ds = R(rhkp, ets)*(s0p-s0)*dzhip;
                    }
                    if (upd)
                    {   // scalar
                        VT t = TPREF ? tcs : tld(pst, bo0);
                        if constexpr (ADV)
                        {
                            t += advec25_hor_f0(sk, s0, TI, u_e, u0, v_n, v0, dxi, dyi);
                            t += vert_combine(otc, obc, Ts, cTs, Gs, cGs, rk, rk1, dzi);
                        }
                        if constexpr (DIF)
                        {
                            const VT e0 = ek[0];
                            const VT ee = div_tpr(e0     +ek[1 ]) + svisc;
                            const VT ew = div_tpr(ek[-1 ]+e0    ) + svisc;
                            const VT en = div_tpr(e0     +ek[TE]) + svisc;
                            const VT es = div_tpr(ek[-TE]+e0    ) + svisc;
                            const VT hor = + ( ee*(sk[1 ]-s0) - ew*(s0-sk[-1 ]) ) * dxidxi
                                           + ( en*(sk[TI]-s0) - es*(s0-sk[-TI]) ) * dyidyi;
                            VT ver;
                            if (fb)      ver = div_rho( Ds + rhk * ld2d(f.sfb, ij), rk, rk1 ) * dzi;
                            else if (ft) ver = div_rho( -rhkp * ld2d(f.sft, ij) - cDs, rk, rk1 ) * dzi;
                            else         ver = div_rho( Ds - cDs, rk, rk1 ) * dzi;
                            t += hor + ver;
                        }
                        if (DSTORE) dss = t; else tst(pst, bo0, t);           // (dsp is set after the sections: no lane-mask value carried through the section loop)
                    }
                }
            }
        };
        mfence(); section(0); mfence(); section(1); mfence(); section(2); mfence(); section(3);
        if (DSTORE) dsp = upd;
        // ---- carry the top faces down, rotate the windows, advance the plane pointers ------------------------------------
        cTu = Tu; cGu = Gu; cDu = Du; cTv = Tv; cGv = Gv; cDv = Dv; cTw = Tw; cGw = Gw; cDw = Dw; cTs = Ts; cGs = Gs; cDs = Ds;
        u1m = u_e; vNm = v_n;
        // (opaque after the add: as recognisable induction variables the compiler keeps the loop-invariant START values and one
        //  running scalar instead -- and under register pressure parks the start values in scratch, a scratch_load + vmcnt(0) per use)
#pragma unroll
        for (int n=0; n<NLD; ++n) { off[n] += kk8w; keep_vgpr(off[n]); }
#pragma unroll
        for (int n=0; n<NLDE; ++n) { offe[n] += kk8w; keep_vgpr(offe[n]); }
        bo0 += kk8w; keep_vgpr(bo0);
        // unconditional (also after the chunk's last level, where nothing is in flight): every path back to the loop head
        // then carries a vmcnt(0) the compiler can see, and it inserts no wait of its own in the next level
        MHH_STAMP(1);                                          // the level's arithmetic, LDS reads, stores
        wait_vmem();                                           // this wave's copies have landed (and its stores have left)
        MHH_STAMP(2);                                          // waited for memory
        __syncthreads();                                       // ... everyone's have, and everyone is done with the oldest planes
        MHH_STAMP(3);                                          // waited for the other waves
        if constexpr (ROT < 0) { shift6(uw, nu); shift6(vw, nv); shift6(ww, nw); shift6(sw, ns); }
        else { uw[RR] = nu; vw[RR] = nv; ww[RR] = nw; sw[RR] = ns; }     // the slot of level k-2 takes level k+4: rotation RR+1
    };
    using std::true_type; using std::false_type;
    using Shift = std::integral_constant<int, -1>;
    bool rho_one = true;                                           // wave-uniform: scalar loads of the chunk's base state
    for (int k = ks; k <= ke; ++k) rho_one = rho_one && (f.rhoref[k] == TF(1.)) && (f.rhorefh[k] == TF(1.));
#ifdef MHH_MARCH_NO_RHO1   // A/B builds only
    rho_one = false;
#endif
    auto chunk = [&](auto rho1_tag, auto unroll_tag) __attribute__((always_inline))
    {
        int k = ks;
        for (; __builtin_expect(k < kf0, 0); ++k) level(k, false_type{}, rho1_tag, Shift{});
        if constexpr (decltype(unroll_tag)::value)
            for (; k + 6 <= kf1; k += 6)
            {
                level(k,   true_type{}, rho1_tag, std::integral_constant<int, 0>{});
                level(k+1, true_type{}, rho1_tag, std::integral_constant<int, 1>{});
                level(k+2, true_type{}, rho1_tag, std::integral_constant<int, 2>{});
                level(k+3, true_type{}, rho1_tag, std::integral_constant<int, 3>{});
                level(k+4, true_type{}, rho1_tag, std::integral_constant<int, 4>{});
                level(k+5, true_type{}, rho1_tag, std::integral_constant<int, 5>{});
            }
        for (; __builtin_expect(k < kf1, 0); ++k) level(k, true_type{}, rho1_tag, Shift{});
        for (; __builtin_expect(k < ke, 0); ++k) level(k, false_type{}, rho1_tag, Shift{});
    };
    if constexpr (sizeof(VT) == 4 && !MHH_MARCH_UNROLL_F32)
    {   // fp32: one body (127 VGPRs, 3-4 waves per SIMD); more bodies cost it a wave
        for (int k = ks; k < ke; ++k) level(k, false_type{}, false_type{}, Shift{});
    }
    else if (__builtin_expect(rho_one, 1)) chunk(true_type{}, true_type{});
    else              chunk(false_type{}, false_type{});       // anelastic base state: shifted windows (one interior body)
    if (DSTORE && dsp && active) tst(pst, bo0 - kk8w, dss);
#ifdef MHH_MARCH_STAMP
    if (tx == 0 && f.dbg) for (int n=0; n<5; ++n) f.dbg[((size_t)blockIdx.x*NJ + ty)*8 + n] = stamp_acc[n];
    if (tx == 0 && f.dbg) f.dbg[((size_t)blockIdx.x*NJ + ty)*8 + 5] = (unsigned long long)(ke - ks);
#endif
}
#ifdef MHH_MARCH_STAMP
static unsigned long long* g_stamp_buf = nullptr; static size_t g_stamp_n = 0;
#endif

// mode 0: advec_2i5 + diff_smag2 (the fused pass); 1: advec_2i5 only (p may be null); 2: diff_smag2 only
// VT = lane value type: double, float, or F2 = two fp32 cells per lane (packed arithmetic; needs an even imax)
template<class VT>
int march_launch(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, hipStream_t st, int mode = 0, int j2 = -1, int j3 = -1)
{
    using TF = typename lane_of<VT>::scalar;
    constexpr int CW = lane_of<VT>::cells;
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
    MarchMetrics<TF> mm;
    mm.visc = TF(f->visc); mm.svisc = has_s ? TF(f->svisc[0]) : TF(0); mf.sm = (p && mode != 1) ? p->surface_model : 0;
    mm.dxi = gd.dxi_t; mm.dyi = gd.dyi_t; mm.dxih = TF(0.5)*gd.dxi_t; mm.dyih = TF(0.5)*gd.dyi_t;
    mm.dxd = gd.dxi_d; mm.dyd = gd.dyi_d; mm.dxd2 = TF(2.)*gd.dxi_d; mm.dyd2 = TF(2.)*gd.dyi_d;
    mm.dxidxi = gd.dxidxi_d; mm.dyidyi = gd.dyidyi_d; mm.quarter = TF(0.25); mm.pad1 = TF(0);
    const TF tPr = p ? TF(p->tPr) : TF(1);
    mm.tPr2 = TF(2.)*tPr; mm.rtPr2 = TF(1.)/mm.tPr2;
    MHH_REQUIRE(mode == 1 || !has_s || known_divisor_ok(mm.tPr2), "tPr must be a positive normal number whose significand is not all ones");
    const bool buoy = mode == 0 && has_s && p->buoyancy == 2 && p->th_for_N2 == 0;
    mf.threfh = buoy ? cp<TF>(p->threfh) : nullptr; mf.grav = buoy ? TF(p->grav) : TF(0);
#ifndef MHH_MARCH_KC
#define MHH_MARCH_KC 128
#endif
    // a strip of a few rows (mhh_rhs_exec_rows on the edge rows) takes short k-chunks: enough blocks to fill the GPU
    int kc = (j0 >= 0 && (j1 - j0 + (j2 >= 0 ? j3 - j2 : 0)) * 4 <= g->jmax) ? 16 : MHH_MARCH_KC;
    { const char* e = getenv("MHH_MARCH_KC_RT"); if (e && atoi(e) >= 8) kc = atoi(e); }       // tuning runs: levels per chunk at run time
    // the lanes address a chunk's planes with 32-bit byte offsets from the chunk's first plane: (kc + 8) planes below 4 GB
    const unsigned long long plane_bytes = (unsigned long long)g->ijcells * sizeof(TF);
    while (kc > 8 && (unsigned long long)(kc + 8) * plane_bytes >= (1ull << 32)) kc /= 2;
    MHH_REQUIRE((unsigned long long)(kc + 8) * plane_bytes < (1ull << 32), "a plane of this grid is too large for the marching kernel's 32-bit lane offsets");
    const MarchTiling t = make_march_tiling(g, NJ, kc, j0, j1, 64*CW, j2, j3);
    const unsigned nblocks = march_blocks(t);
#ifdef MHH_MARCH_STAMP
    if (g_stamp_n < (size_t)nblocks*NJ*8) { if (g_stamp_buf) (void)hipFree(g_stamp_buf); g_stamp_n = (size_t)nblocks*NJ*8; MHH_HIP_TRY(hipMalloc(&g_stamp_buf, g_stamp_n*8)); }
    MHH_HIP_TRY(hipMemsetAsync(g_stamp_buf, 0, g_stamp_n*8, st));
    mf.dbg = g_stamp_buf;
#endif
    // 16-byte LDS-DMA needs 16-byte aligned plane rows; other layouts copy in 4-byte pieces (MHH_MARCH_DMA=4 forces that
    // form; same arithmetic in both)
    constexpr int VEC = 16 / (int)sizeof(TF);
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    const char* env = getenv("MHH_MARCH_DMA");
    const bool aligned = (g->icells % VEC == 0) && al16(f->u) && al16(f->v) && al16(f->w) && (mode == 1 || al16(f->evisc)) && (!has_s || al16(f->s[0]));
    const int pb = ((env && !strcmp(env, "4")) || !aligned) ? 4 : 16;
#define MHH_LAUNCH_MARCH(PBV, A, D) do { \
        if (has_s) hipLaunchKernelGGL((rhs25_march_kernel<VT, NJ, true, PBV, A, D>),  dim3(nblocks), dim3(64, NJ), 0, st, mm, gd, mf, t); \
        else       hipLaunchKernelGGL((rhs25_march_kernel<VT, NJ, false, PBV, A, D>), dim3(nblocks), dim3(64, NJ), 0, st, mm, gd, mf, t); } while (0)
    // tile origin on a 16-byte piece where three cells west of the first cell is not one (istart = 16: rows of whole cache lines)
    const char* ehx = getenv("MHH_MARCH_HX");
    const bool hx4 = mode == 0 && has_s && pb == 16 && (g->istart - 3) % VEC != 0 && (g->istart - 4) % VEC == 0 && g->istart >= 4 && !(ehx && !strcmp(ehx, "3"));
    if (hx4) hipLaunchKernelGGL((rhs25_march_kernel<VT, NJ, true, 16, true, true, 4>), dim3(nblocks), dim3(64, NJ), 0, st, mm, gd, mf, t);
    else if (mode == 0) { if (pb == 16) MHH_LAUNCH_MARCH(16, true, true);  else MHH_LAUNCH_MARCH(4, true, true); }
    else if (mode == 1) { if (pb == 16) MHH_LAUNCH_MARCH(16, true, false); else MHH_LAUNCH_MARCH(4, true, false); }
    else                { if (pb == 16) MHH_LAUNCH_MARCH(16, false, true); else MHH_LAUNCH_MARCH(4, false, true); }
#undef MHH_LAUNCH_MARCH
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
} // namespace

#ifdef MHH_MARCH_STAMP
// probe builds: the stamps of the last launch, [block][wave][8] = cycles issuing | computing | waiting for memory | at the barrier | loop control, levels
extern "C" __attribute__((visibility("default"))) long long mhh_march_stamps(unsigned long long* host, long long max_n)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    const long long n = (long long)g_stamp_n < max_n ? (long long)g_stamp_n : max_n;
    if (n > 0 && hipMemcpy(host, g_stamp_buf, (size_t)n*8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
#endif
// fp32: two cells per lane with packed arithmetic where the rows are whole 128-cell tiles; MHH_MARCH_F32X2=0 keeps one cell per lane
static bool f32x2(const mhh_grid* g)
{
    const char* e = getenv("MHH_MARCH_F32X2");
    return g->imax % 128 == 0 && !(e && !strcmp(e, "0"));
}
static int march_dispatch(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, void* stream, int mode, int j2 = -1, int j3 = -1)
{
    if (g->dtype == MHH_F64) return march_launch<double>(g, f, p, j0, j1, as_stream(stream), mode, j2, j3);
    if (f32x2(g)) return march_launch<F2>(g, f, p, j0, j1, as_stream(stream), mode, j2, j3);
    return march_launch<float>(g, f, p, j0, j1, as_stream(stream), mode, j2, j3);
}
// entry used by mhh_rhs_exec for the (advec_2i5, diff_smag2) pair: u, v, w and scalar 0 (inputs validated by the caller)
int mhh_rhs25_march(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, void* stream) { return march_dispatch(g, f, p, -1, -1, stream, 0); }
// the same over the rows [j0, j1) only (interior rows while the halos travel, edge rows after: mhh_rhs_exec_rows)
int mhh_rhs25_march_rows(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, void* stream) { return march_dispatch(g, f, p, j0, j1, stream, 0); }
// two row ranges in one launch (the two edge strips of a slab)
int mhh_rhs25_march_rows2(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, int j2, int j3, void* stream) { return march_dispatch(g, f, p, j0, j1, stream, 0, j2, j3); }

// Advec_2i5::exec / Diff_smag2::exec on their own, for u, v, w and scalar 0 (inputs validated by the caller): the marching
// kernel with one operator's terms only -- what the two calls of an unfused time step run.
int mhh_advec25_march(const mhh_grid* g, const mhh_fields* f, void* stream) { return march_dispatch(g, f, nullptr, -1, -1, stream, 1); }
int mhh_diff_smag2_march(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, void* stream) { return march_dispatch(g, f, p, -1, -1, stream, 2); }
