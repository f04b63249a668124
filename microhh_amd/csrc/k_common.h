// k_common.h -- host-side plumbing shared by the HIP translation units of libmhh_hip.so:
// error reporting, mhh_grid -> GridDev<TF> narrowing, the generic cell-kernel launcher.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/mhh_hip.h"
#include "cell_ops.h"

#define MHH_API extern "C" __attribute__((visibility("default")))

namespace mhh
{
void set_error(const char* fmt, ...);

#define MHH_HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    mhh::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return MHH_EHIP; } } while (0)
#define MHH_REQUIRE(cond, msg) do { if (!(cond)) { mhh::set_error("%s: requirement `%s` failed (%s)", __func__, #cond, msg); return MHH_EINVAL; } } while (0)
#define MHH_LAUNCH_CHECK() MHH_HIP_TRY(hipGetLastError())

inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

template<class TF> inline const TF* cp(const void* p) { return static_cast<const TF*>(p); }
template<class TF> inline TF* mp(void* p) { return static_cast<TF*>(p); }

// Narrow the ABI grid to the kernel-side descriptor, spelling each reciprocal the way the reference's
// call sites do (see GridDev comments).
template<class TF>
inline GridDev<TF> make_grid(const mhh_grid* g)
{
    GridDev<TF> d;
    d.itot = g->itot; d.jtot = g->jtot; d.ktot = g->ktot; d.imax = g->imax; d.jmax = g->jmax; d.kmax = g->kmax;
    d.igc = g->igc; d.jgc = g->jgc; d.kgc = g->kgc;
    d.icells = g->icells; d.jcells = g->jcells; d.ijcells = g->ijcells; d.kcells = g->kcells;
    d.istart = g->istart; d.iend = g->iend; d.jstart = g->jstart; d.jend = g->jend; d.kstart = g->kstart; d.kend = g->kend;
    d.dim3 = (g->jtot != 1);
    const TF dx = TF(g->dx), dy = TF(g->dy);
    d.dx = dx; d.dy = dy; d.zsize = TF(g->zsize);
    d.dxi_t = TF(1.)/dx;       d.dyi_t = TF(1.)/dy;
    d.dxi_d = TF(1./dx);       d.dyi_d = TF(1./dy);
    d.dxidxi_d = TF(1./(dx*dx)); d.dyidyi_d = TF(1./(dy*dy));
    d.dxidxi_t = TF(1/(dx*dx));  d.dyidyi_t = TF(1/(dy*dy));
    d.dxidxi_2 = 1/(dx*dx);      d.dyidyi_2 = 1/(dy*dy);
    d.z = cp<TF>(g->z); d.dz = cp<TF>(g->dz); d.dzi = cp<TF>(g->dzi); d.dzhi = cp<TF>(g->dzhi);
    d.dzi4 = cp<TF>(g->dzi4); d.dzhi4 = cp<TF>(g->dzhi4);
    return d;
}

inline int check_grid(const mhh_grid* g)
{
    if (!g) { set_error("null grid"); return MHH_EINVAL; }
    if (g->dtype != MHH_F64 && g->dtype != MHH_F32) { set_error("grid dtype must be MHH_F64 or MHH_F32"); return MHH_EINVAL; }
    if (g->icells != g->imax + 2*g->igc || g->jcells != g->jmax + 2*g->jgc || g->kcells != g->kmax + 2*g->kgc ||
        g->ijcells != g->icells*g->jcells || g->iend - g->istart != g->imax || g->jend - g->jstart != g->jmax ||
        g->kend - g->kstart != g->kmax || g->istart != g->igc || g->jstart != g->jgc || g->kstart != g->kgc ||
        g->imax < 1 || g->jmax < 1 || g->kmax < 1)
    { set_error("inconsistent grid index bundle"); return MHH_EINVAL; }
    if (g->npx != 1) { set_error("only slab decomposition (npx == 1) is supported"); return MHH_EINVAL; }
    // cell indices are 32-bit ints, as the reference's ijk (include/grid.h): refuse grids whose ghosted size does not fit,
    // with room for the stencil offsets, instead of overflowing (2^31 cells x 8 B = 17 GB per field: possible on 288 GB)
    if ((long long)g->ijcells * (long long)(g->kcells + 4) >= 2147483647LL || (long long)g->icells * (long long)g->jcells >= 2147483647LL)
    { set_error("grid too large for 32-bit cell indices (icells*jcells*(kcells+4) must stay below 2^31)"); return MHH_EINVAL; }
    return MHH_OK;
}

// ---- generic one-thread-per-cell kernel --------------------------------------------------------------
// Block = BX x BY threads over (i, j); blockIdx.z walks the k range, so everything that depends on k
// (metric rows, density, 2i5 face orders, wall branches) is wave-uniform.
#ifndef MHH_BX
#define MHH_BX 64
#define MHH_BY 4
#endif
constexpr int BX = MHH_BX, BY = MHH_BY;    // BX must stay a multiple of 64 (wave_reduce.h assumes BX == 64 for reductions)

// Block -> tile mapping, XCD-aware. MI355X has 8 XCDs with a private 4 MiB L2 each and workgroups are dealt
// round-robin over them (block L runs on XCD L % 8: observed placement, used for speed only -- any other
// placement gives the same results). A stencil sweep re-reads every plane 3..7 times in k and every row 3..7
// times in j; dealing neighbouring tiles to different XCDs makes each L2 hold the whole working set and miss
// (measured: 6x the algorithmic HBM bytes on the fused 2i5+smag2 pass at 512^3). So each XCD gets whole
// "strips" of SR block-rows (all i), and walks a strip plane by plane in k before taking its next strip:
// the live working set of an XCD is (rows of a strip + halo) x (planes of the stencil + planes in flight).
struct Tiling { int nbx, nby, nk, sr, ns, nseg, kseg; };
#ifndef MHH_STRIP_ROWS
#define MHH_STRIP_ROWS 16          // grid rows per strip (tuned on MI355X)
#endif

// A "unit" is one strip (sr block-rows, all i) over one k-segment; units are dealt to the XCDs round-robin and a unit
// is walked plane by plane. nseg > 1 only when there are fewer than 8 strips (thin slabs of a multi-GPU run), so
// that all 8 XCDs still get work.
__device__ __forceinline__ bool decode_tile(const Tiling& t, unsigned L, int& bx, int& by, int& kz)
{
    const int xcd = L & 7u;
    const unsigned tt = L >> 3;
    const unsigned per_unit = (unsigned)t.sr * t.nbx * t.kseg;
    const unsigned round = tt / per_unit;
    unsigned r = tt - round * per_unit;
    const int unit = (int)round * 8 + xcd;
    if (unit >= t.ns * t.nseg) return false;
    const int strip = unit % t.ns, seg = unit / t.ns;
    const unsigned per_plane = (unsigned)t.sr * t.nbx;
    const int kl = (int)(r / per_plane); r -= (unsigned)kl * per_plane;
    const int byl = (int)(r / t.nbx);
    bx = (int)(r - (unsigned)byl * t.nbx);
    by = strip * t.sr + byl;
    kz = seg * t.kseg + kl;
    return (by < t.nby) && (kz < t.nk);
}

template<class Op>
__global__ void __launch_bounds__(BX*BY) cell_kernel(const Op op, const Tiling t, int i0, int i1, int j0, int j1, int k0, int jj, int kk)
{
    int bx, by, kz;
    if (!decode_tile(t, blockIdx.x, bx, by, kz)) return;
    const int i = i0 + bx*BX + threadIdx.x;
    const int j = j0 + by*BY + threadIdx.y;
    const int k = k0 + kz;
    if (i < i1 && j < j1)
        op(i, j, k, i + j*jj + k*kk);
}

inline Tiling make_tiling(int ni, int nj, int nk)
{
    Tiling t;
    t.nbx = (ni + BX-1)/BX; t.nby = (nj + BY-1)/BY; t.nk = nk;
    t.sr = (MHH_STRIP_ROWS + BY-1)/BY;
    if (t.sr > t.nby/8) t.sr = t.nby/8;          // at least 8 strips when the slab is tall enough
    if (t.sr < 1) t.sr = 1;
    t.ns = (t.nby + t.sr-1)/t.sr;
    t.nseg = (t.ns >= 8) ? 1 : (8 + t.ns-1)/t.ns;
    if (t.nseg > nk) t.nseg = nk;
    t.kseg = (nk + t.nseg-1)/t.nseg;
    return t;
}
inline unsigned tiling_blocks(const Tiling& t) { const int units = t.ns*t.nseg; return 8u * (unsigned)((units + 7)/8) * (unsigned)t.sr * t.nbx * t.kseg; }

template<class Op>
inline int launch_cells(hipStream_t st, const Op& op, int i0, int i1, int j0, int j1, int k0, int k1, int jj, int kk)
{
    if (k1 <= k0 || i1 <= i0 || j1 <= j0) return MHH_OK;
    const Tiling t = make_tiling(i1-i0, j1-j0, k1-k0);
    hipLaunchKernelGGL(cell_kernel<Op>, dim3(tiling_blocks(t)), dim3(BX, BY, 1), 0, st, op, t, i0, i1, j0, j1, k0, jj, kk);
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
template<class TF, class Op>
inline int launch_interior(hipStream_t st, const GridDev<TF>& g, int k0, int k1, const Op& op)
{
    return launch_cells(st, op, g.istart, g.iend, g.jstart, g.jend, k0, k1, g.icells, g.ijcells);
}

#define MHH_DISPATCH(g, CALL) ((g)->dtype == MHH_F64 ? CALL(double) : CALL(float))

// The inverse transform's normalisation, value / jtot / itot (src/fft.cxx). Where both extents are powers of two (POW2) the
// two divisions are two multiplications by the exact reciprocals: the same correctly rounded results, a tenth of the issue slots.
template<bool POW2, class TF>
__device__ __forceinline__ TF fft_norm(TF v, int itot, int jtot, TF ri, TF rj) { return POW2 ? (v * rj) * ri : v / jtot / itot; }
static inline bool is_pow2(int n) { return n > 0 && (n & (n-1)) == 0; }


// a divisor div_known (cell_ops.h) may take: positive, normal, significand not all ones (1/3, 2/3, 1, 0.7 ... qualify)
template<class TF> inline bool known_divisor_ok(TF d)
{
    if (!(d > TF(0)) || !std::isfinite(d) || !std::isnormal(d)) return false;
    int e; const TF m = std::frexp(d, &e);                          // m in [0.5, 1)
    return std::nextafter(m, TF(1)) != TF(1);
}

// Low-storage Runge-Kutta coefficients of the time loop (src/timeloop.cxx:250-334): a += cB[substep]*dt*at, then at *= cA[next]
// (or at = 0 over all cells when the next sub-step is the first of a new step)
inline bool rk_coefficients(int rkorder, int substep, double& cA, double& cB, bool& reset)
{
    static const double A3[] = {0., -5./9., -153./128.};
    static const double B3[] = {1./3., 15./16., 8./15.};
    static const double A4[] = {0., -567301805773./1357537059087., -2404267990393./2016746695238., -3550918686646./2091501179385., -1275806237668./842570457699.};
    static const double B4[] = {1432997174477./9575080441755., 5161836677717./13612068292357., 1720146321549./2090206949498., 3134564353537./4481467310338., 2277821191437./14882151754819.};
    if (rkorder != 3 && rkorder != 4) return false;
    const int ns = (rkorder == 3) ? 3 : 5;
    if (substep < 0 || substep >= ns) return false;
    const int nxt = (substep+1) % ns;
    cA = (rkorder == 3) ? A3[nxt] : A4[nxt]; cB = (rkorder == 3) ? B3[substep] : B4[substep];
    reset = (nxt == 0);
    return true;
}
} // namespace mhh
