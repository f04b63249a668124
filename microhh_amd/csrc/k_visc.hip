// k_visc.hip -- Diff_smag2::exec_viscosity (strain^2 -> N2 -> evisc, src/diff_smag2.cxx:47-155,254-367) as a k-marching
// LDS kernel for gfx950, the companion of k_march.hip.
//
// The one-thread-per-cell form (ViscosityOp, k_rhs.hip) issues 26 vector loads per cell, most of them one element off a
// 512-byte row, and is bound by the L1/TA pipe (2.25 ms at 512^3 fp64 for 40 B/cell of algorithmic traffic). Here a
// 64 x NJ block walks up in k: the planes k and k+1 of u, v, w live in LDS with a one-cell halo (LDS-DMA, one barrier
// per level), and the four vertical-shear terms on the top face of level k are carried to level k+1 as its bottom-face
// terms -- the two expressions have the same operands in the same order (src/diff_smag2.cxx:138-148), so the sum is the
// reference's, bit for bit.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include "k_common.h"
#include "k_march_common.h"
#include <gfx950_prims.h>

using namespace mhh;

#ifdef MHH_FMA_BUILD     // the named FMA build (build.py): its kernels carry their own name in profiler output
#define visc_march_kernel visc_march_fma_kernel
#endif

namespace
{
template<class TF> struct ViscFields
{
    const TF* __restrict__ u; const TF* __restrict__ v; const TF* __restrict__ w; TF* __restrict__ ev;
    const TF* __restrict__ dudz; const TF* __restrict__ dvdz; const TF* __restrict__ dbdz; const TF* __restrict__ z0m;
    const TF* __restrict__ N2; const TF* __restrict__ th; const TF* __restrict__ thref; const TF* __restrict__ mlen0; const TF* __restrict__ mlen2;
    TF grav, tPr; int sm, neutral, ex;
    TF rtPr;                       // RN(1 / tPr) from the host, or 0: the division by tPr as it stands (div_known, cell_ops.h)
};

// acc + 0.125*q and 2*acc + c as ONE fma each: the products with powers of two are exact (q, acc far from the subnormal range or exactly
// zero), so the fma rounds where the reference's addition rounds -- the same bits, a vector instruction less per term. MHH_VISC_EXACT_FMA=0: as written.
#ifndef MHH_VISC_EXACT_FMA
#define MHH_VISC_EXACT_FMA 1
#endif
template<class TF> __device__ __forceinline__ TF add_eighth(TF acc, TF q) { return MHH_VISC_EXACT_FMA ? tfma(TF(0.125), q, acc) : acc + TF(0.125)*q; }

#ifndef MHH_VISC_OCC
#define MHH_VISC_OCC 4
#endif
#ifndef MHH_VISC_KC
#define MHH_VISC_KC 64
#endif
// EXT_N2: N2 comes from a caller-supplied 3-D field instead of th. Its own instantiation, so that the usual form has no
// vector load (and hence no compiler-placed s_waitcnt vmcnt(0), which would also wait for the plane copies) inside a level.
template<class TF, int NJ, int PB, bool EXT_N2>
__global__ void __launch_bounds__(64*NJ, MHH_VISC_OCC) visc_march_kernel(const GridDev<TF> g, const ViscFields<TF> f, const MarchTiling mt)
{
    constexpr int TI = 72, TJ = NJ + 2, NT = 64*NJ, NTILE = TI*TJ;    // tile x from i0-ex (ex <= 4: 64 + ex + 1 <= 72)
    constexpr int R = 3;                                              // ring: planes k, k+1 and the copy in flight
    __shared__ __attribute__((aligned(16))) TF U[R][NTILE];
    __shared__ __attribute__((aligned(16))) TF V[R][NTILE];
    __shared__ __attribute__((aligned(16))) TF W[R][NTILE];

    int bx, by, kcn;
    if (!decode_march(mt, blockIdx.x, bx, by, kcn)) return;
    const int jj = g.icells, kk = g.ijcells;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty*64 + tx;
    int j0, jlim; march_tile_rows(mt, by, NJ, j0, jlim);
    const int i0 = g.istart + bx*64;
    const int kb = g.kstart + kcn*mt.kc;
    const int ke = (kb + mt.kc < g.kend) ? kb + mt.kc : g.kend;
    const int i = i0 + tx, j = j0 + ty;
    const bool active = (i < g.iend) && (j < jlim);
    const int ci = (i < g.iend) ? i : g.iend-1, cj = (j < jlim) ? j : jlim-1;
    const int col = ci + cj*jj, ij = col;
    const int l = (ty+1)*TI + (tx+f.ex);
    auto slot = [](int p) { return (p + 12) % R; };

    TileCopy<TF, PB, TI, TJ, NT> tc;
    tc.init(tid, i0 - f.ex, j0 - 1, g.icells, g.jcells);
    auto dma_tile = [&](const TF* __restrict__ fld, int kp, TF* __restrict__ lds)
    {
        if (kp < 0 || kp >= g.kcells) return;                         // wave-uniform
        tc.copy(fld + (size_t)kp*kk, lds);
    };
    auto colth = [&](int kp) -> TF { return (f.th && kp >= 0 && kp < g.kcells) ? f.th[col + kp*kk] : TF(0); };

    // ---- prologue: planes ks and ks+1; ks = kb-1 is the warm-up level (only top-face terms are formed there) ----------
    const int ks = kb - 1;
    for (int p = ks; p <= ks+1; ++p)
    {
        dma_tile(f.u, p, U[slot(p)]); dma_tile(f.v, p, V[slot(p)]); dma_tile(f.w, p, W[slot(p)]);
    }
    TF thm = colth(ks-1), thc = colth(ks), thp = colth(ks+1);         // th at k-1, k, k+1 of this column
    // the 2-D surface inputs of this column, once, ahead of the prologue's wait: a load inside the loop is followed by an
    // s_waitcnt vmcnt(0) at its use, which would also wait for the plane copies just issued
    const bool surf = f.sm && (kb == g.kstart);
    const TF z0m_c = f.sm ? f.z0m[ij] : TF(0);
    const TF dudz_c = surf ? f.dudz[ij] : TF(0), dvdz_c = surf ? f.dvdz[ij] : TF(0), dbdz_c = (surf && !f.neutral) ? f.dbdz[ij] : TF(0);
    wait_vmem();
    __syncthreads();

    // grav / thref[k] is a per-level constant, but an fp64 division has no scalar form: lane l of every wave divides for level kb + l
    // ONCE per chunk (chunks are at most 64 levels), and a level takes its quotient from that lane -- the same correctly rounded
    // quotient, ~12 vector instructions per level less
#ifndef MHH_VISC_SQRT_RANGE
#define MHH_VISC_SQRT_RANGE 1
#endif
#ifndef MHH_VISC_LANE_QUOT
#define MHH_VISC_LANE_QUOT 1
#endif
    const bool lane_quot = MHH_VISC_LANE_QUOT && !f.neutral && !EXT_N2 && f.thref && (ke - kb) <= 64;
    TF gq = TF(0);
    if (lane_quot) { const int kq = kb + tx; gq = f.grav / f.thref[kq < g.kcells ? kq : g.kcells-1] * TF(0.5); }   // grav/thref[k] * 0.5: the first product of N2 below
    TF bu0 = 0, bu1 = 0, bv0 = 0, bv1 = 0;                            // carried bottom-face terms: the squares (their 1/8 is applied where they are added)
    // The result of level k is stored at the top of iteration k+1, after the next copy has been issued: s_waitcnt vmcnt(0)
    // before the barrier also waits for stores (gfx9 has one counter), and a store issued right before it would expose its
    // full latency on every level.
    TF ev_pending = 0; int c_pending = -1;
    const TF dxi = g.dxi_d, dyi = g.dyi_d;

    for (int k = ks; k < ke; ++k)
    {
        const bool more = (k + 1 < ke);
        if (more) { dma_tile(f.u, k+2, U[slot(k+2)]); dma_tile(f.v, k+2, V[slot(k+2)]); dma_tile(f.w, k+2, W[slot(k+2)]); }
        const TF thn = more ? colth(k+2) : TF(0);
        if (c_pending >= 0) { f.ev[c_pending] = ev_pending; c_pending = -1; }

        const TF* __restrict__ uk = U[slot(k)] + l;  const TF* __restrict__ ukp = U[slot(k+1)] + l;
        const TF* __restrict__ vk = V[slot(k)] + l;  const TF* __restrict__ vkp = V[slot(k+1)] + l;
        const TF* __restrict__ wk = W[slot(k)] + l;  const TF* __restrict__ wkp = W[slot(k+1)] + l;
        const TF dzhip = uniform_load(g.dzhi, k+1);

        // top-face shear terms of level k == bottom-face terms of level k+1 (src/diff_smag2.cxx:140-141,146-147 vs :138-139,144-145)
        const TF tu0 = sq((ukp[0 ]-uk[0 ])*dzhip + (wkp[0 ]-wkp[-1 ])*dxi);
        const TF tu1 = sq((ukp[1 ]-uk[1 ])*dzhip + (wkp[1 ]-wkp[0  ])*dxi);
        const TF tv0 = sq((vkp[0 ]-vk[0 ])*dzhip + (wkp[0 ]-wkp[-TI])*dyi);
        const TF tv1 = sq((vkp[TI]-vk[TI])*dzhip + (wkp[TI]-wkp[0  ])*dyi);

        if (k >= kb && active)
        {
            const bool mo = f.sm && (k == g.kstart);
            TF acc = sq((uk[1]-uk[0])*dxi);
            acc = acc + sq((vk[TI]-vk[0])*dyi);
            acc = acc + sq((wkp[0]-wk[0])*uniform_load(g.dzi, k));
            acc = add_eighth(acc, sq((uk[0    ]-uk[-TI  ])*dyi + (vk[0   ]-vk[-1   ])*dxi));
            acc = add_eighth(acc, sq((uk[1    ]-uk[1-TI ])*dyi + (vk[1   ]-vk[0    ])*dxi));
            acc = add_eighth(acc, sq((uk[TI   ]-uk[0    ])*dyi + (vk[TI  ]-vk[TI-1 ])*dxi));
            acc = add_eighth(acc, sq((uk[1+TI ]-uk[1    ])*dyi + (vk[1+TI]-vk[TI   ])*dxi));
            if (mo)
            {   // unresolved wall: MOST gradients replace the resolved du/dz, dv/dz (src/diff_smag2.cxx:72-114)
                acc = acc + TF(0.5)*sq(dudz_c);
                acc = acc + TF(0.125)*sq((wk [0 ]-wk [-1 ])*dxi);
                acc = acc + TF(0.125)*sq((wk [1 ]-wk [0  ])*dxi);
                acc = acc + TF(0.125)*sq((wkp[0 ]-wkp[-1 ])*dxi);
                acc = acc + TF(0.125)*sq((wkp[1 ]-wkp[0  ])*dxi);
                acc = acc + TF(0.5)*sq(dvdz_c);
                acc = acc + TF(0.125)*sq((wk [0 ]-wk [-TI])*dyi);
                acc = acc + TF(0.125)*sq((wk [TI]-wk [0  ])*dyi);
                acc = acc + TF(0.125)*sq((wkp[0 ]-wkp[-TI])*dyi);
                acc = acc + TF(0.125)*sq((wkp[TI]-wkp[0  ])*dyi);
            }
            else
            {
                acc = add_eighth(acc, bu0); acc = add_eighth(acc, bu1); acc = add_eighth(acc, tu0); acc = add_eighth(acc, tu1);
                acc = add_eighth(acc, bv0); acc = add_eighth(acc, bv1); acc = add_eighth(acc, tv0); acc = add_eighth(acc, tv1);
            }
            TF s2;
            if (MHH_VISC_EXACT_FMA) s2 = tfma(TF(2.), acc, TF(1.e-9));      // 2*acc is exact: one rounding, where `s2 += 1e-9` rounds
            else { s2 = TF(2.)*acc; s2 += 1.e-9; }
            const int c = col + k*kk;
            TF n2 = TF(0);
            if (!f.neutral)
            {
                if (mo) n2 = dbdz_c;
                else if (EXT_N2) n2 = f.N2[c];
                else
                {
                    const TF gth = lane_quot ? value_of_lane(gq, k - kb, f.grav/f.thref[k]*TF(0.5)) : f.grav/uniform_load(f.thref, k)*TF(0.5);
                    n2 = gth*(thp - thm)*uniform_load(g.dzi, k);
                }
            }
            c_pending = c;
            const TF fac = f.mlen2 ? uniform_load(f.mlen2, k)            // uniform z0m: per-level table, same bits
                                   : evisc_mlen2(f.sm, f.neutral, uniform_load(f.mlen0, k), f.sm ? uniform_load(g.z, k) : TF(0), z0m_c);
            // evisc_from_mlen2 (cell_ops.h) with its two square roots as sqrt_in_range (s2 >= 1e-9, 1 - rit >= 1e-9) and, where tPr
            // allows, its division by the uniform tPr as div_known (the correctly rounded quotient in five fmas)
            const TF rs2 = MHH_VISC_SQRT_RANGE ? sqrt_in_range(s2) : dsqrt2(s2);
            if (f.neutral) ev_pending = fac * rs2;
            else
            {
                TF rit = (f.rtPr != TF(0)) ? div_known(n2 / s2, f.tPr, f.rtPr) : n2 / s2 / f.tPr;
                rit = tmin(rit, TF(1.-1.e-9));
                const TF om = TF(1.)-rit;
                ev_pending = fac * rs2 * (MHH_VISC_SQRT_RANGE ? sqrt_in_range(om) : dsqrt2(om));
            }
        }
        bu0 = tu0; bu1 = tu1; bv0 = tv0; bv1 = tv1;
        wait_vmem();                  // unconditional: every path back to the loop head carries a vmcnt(0) the compiler can see
        __syncthreads();
        if (more) { thm = thc; thc = thp; thp = thn; }
    }
    if (c_pending >= 0) f.ev[c_pending] = ev_pending;
}

template<class TF>
int visc_launch(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, const void* th, int ex, int pb, int j0, int j1, hipStream_t st, int j2 = -1, int j3 = -1)
{
    constexpr int NJ = 4;
    ViscFields<TF> vf;
    vf.u = cp<TF>(f->u); vf.v = cp<TF>(f->v); vf.w = cp<TF>(f->w); vf.ev = mp<TF>(f->evisc);
    vf.dudz = cp<TF>(f->dudz); vf.dvdz = cp<TF>(f->dvdz); vf.dbdz = cp<TF>(f->dbdz); vf.z0m = cp<TF>(f->z0m);
    vf.N2 = cp<TF>(p->N2); vf.th = cp<TF>(th); vf.thref = cp<TF>(p->thref); vf.mlen0 = cp<TF>(p->mlen0); vf.mlen2 = cp<TF>(p->mlen2);
    vf.grav = TF(p->grav); vf.tPr = TF(p->tPr); vf.sm = p->surface_model; vf.neutral = p->neutral; vf.ex = ex;
#ifndef MHH_VISC_DIVKNOWN
#define MHH_VISC_DIVKNOWN 1
#endif
    vf.rtPr = (MHH_VISC_DIVKNOWN && known_divisor_ok(vf.tPr)) ? TF(1.)/vf.tPr : TF(0);
    int kc = (j0 >= 0 && (j1 - j0 + (j2 >= 0 ? j3 - j2 : 0)) * 4 <= g->jmax) ? 16 : MHH_VISC_KC;      // few rows: short k-chunks fill the GPU
    { const char* e = getenv("MHH_VISC_KC_RT"); if (e && atoi(e) >= 8) kc = atoi(e); }         // tuning runs
    const MarchTiling t = make_march_tiling(g, NJ, kc, j0, j1, 64, j2, j3);
    const dim3 nb(march_blocks(t)), bs(64, NJ);
    const GridDev<TF> gd = make_grid<TF>(g);
    if (vf.N2)
    {
        if (pb == 16) hipLaunchKernelGGL((visc_march_kernel<TF, NJ, 16, true>), nb, bs, 0, st, gd, vf, t);
        else          hipLaunchKernelGGL((visc_march_kernel<TF, NJ, 4, true>),  nb, bs, 0, st, gd, vf, t);
    }
    else
    {
        if (pb == 16) hipLaunchKernelGGL((visc_march_kernel<TF, NJ, 16, false>), nb, bs, 0, st, gd, vf, t);
        else          hipLaunchKernelGGL((visc_march_kernel<TF, NJ, 4, false>),  nb, bs, 0, st, gd, vf, t);
    }
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
} // namespace

// ---- self test: sqrt_in_range (gfx950_prims.h) against the compiler's sqrt, bit for bit ------------------------------------
namespace
{
__global__ void __launch_bounds__(256) sqrt_selftest_kernel(unsigned long long n, unsigned long long seed, unsigned long long* bad)
{
    unsigned long long miss = 0;
    for (unsigned long long i = blockIdx.x*256ull + threadIdx.x; i < n; i += gridDim.x*256ull)
    {
        // splitmix64 of (seed, i): a random significand under an exponent in [2^-767, 2^1023]; every 16th value near the lower
        // bounds the kernel feeds it (1e-9 and a few ulps above)
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull*(i+1);
        z = (z ^ (z >> 30))*0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27))*0x94D049BB133111EBull; z ^= z >> 31;
        const unsigned long long mant = z & 0xFFFFFFFFFFFFFull;
        const unsigned long long e = 256ull + (z >> 52) % (2046ull - 256ull + 1ull);        // biased exponent 256 (2^-767) ... 2046
        double x = __builtin_bit_cast(double, (e << 52) | mant);
        if ((i & 15u) == 15u) x = __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, 1.e-9) + (z & 1023u));
        const double a = sqrt_in_range(x), b = __builtin_sqrt(x);
        miss += (__builtin_bit_cast(unsigned long long, a) != __builtin_bit_cast(unsigned long long, b));
    }
    if (miss) atomicAdd(bad, miss);
}
}
// counts the arguments (n of them, drawn from `seed`) on which sqrt_in_range and sqrt differ: *mismatches must come back 0
MHH_API int mhh_selftest_sqrt_in_range(unsigned long long n, unsigned long long seed, unsigned long long* mismatches, void* stream)
{
    MHH_REQUIRE(mismatches != nullptr, "null result pointer");
    unsigned long long* d = nullptr;
    MHH_HIP_TRY(hipMalloc((void**)&d, sizeof(*d)));
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemsetAsync(d, 0, sizeof(*d), st);
    if (e == hipSuccess) { hipLaunchKernelGGL(sqrt_selftest_kernel, dim3(2048), dim3(256), 0, st, n, seed, d); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(mismatches, d, sizeof(*d), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) { mhh::set_error("sqrt self test failed: %s", hipGetErrorString(e)); return MHH_EHIP; }
    return MHH_OK;
}

static unsigned long long g_visc_march_launches = 0;
// diagnostics: how many times the marching form (as opposed to the one-thread-per-cell form) has been launched
MHH_API unsigned long long mhh_stat_visc_march_launches(void) { return g_visc_march_launches; }

// Entry used by mhh_diff_exec_viscosity (inputs validated there). Returns 1 when the marching kernel ran, 0 when it is
// switched off (MHH_VISC_IMPL=cell) or the grid has no ghost cells to read (the caller then takes the cell kernel),
// < 0 on error (-code).
int mhh_visc_march_rows2(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, const void* th, int j0, int j1, int j2, int j3, void* stream);
int mhh_visc_march_rows(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, const void* th, int j0, int j1, void* stream)
{ return mhh_visc_march_rows2(g, f, p, th, j0, j1, -1, -1, stream); }
int mhh_visc_march(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, const void* th, void* stream)
{ return mhh_visc_march_rows(g, f, p, th, -1, -1, stream); }
// rows [j0, j1) only (-1, -1: the interior) and optionally [j2, j3) in the same launch; ghost rows jstart-1 and jend are legal when jgc >= 2
int mhh_visc_march_rows2(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, const void* th, int j0, int j1, int j2, int j3, void* stream)
{
    { const char* e = getenv("MHH_VISC_IMPL"); if (e && !strcmp(e, "cell")) return 0; }     // A/B switch, read per call
    const int vec = (g->dtype == MHH_F64) ? 2 : 4;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    if (g->igc < 1 || g->jgc < 1 || g->kgc < 1) return 0;
    // 16-byte pieces: tile x-origin i0 - ex (i0 = igc + 64*bx) on a 16-byte boundary, covering i-1: 1 <= ex <= igc,
    // (igc - ex) % vec == 0; any other layout copies in 4-byte pieces with ex = 1
    int ex = 0;
    if (g->icells % vec == 0 && al16(f->u) && al16(f->v) && al16(f->w))
        for (int e = 1; e <= g->igc && e <= 4; ++e) if ((g->igc - e) % vec == 0) { ex = e; break; }
    const int pb = ex ? 16 : 4;
    if (!ex) ex = 1;
    ++g_visc_march_launches;
    const int rc = (g->dtype == MHH_F64) ? visc_launch<double>(g, f, p, th, ex, pb, j0, j1, as_stream(stream), j2, j3) : visc_launch<float>(g, f, p, th, ex, pb, j0, j1, as_stream(stream), j2, j3);
    return rc == MHH_OK ? 1 : -rc;
}
