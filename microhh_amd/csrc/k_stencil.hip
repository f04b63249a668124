// k_stencil.hip -- one-thread-per-cell HIP kernels of the stencil operators at the reference's kernel
// granularity (advec_u/v/w/s, diff_c/w, smag2 pieces, boundary_cyclic, calc_N2, rk substep) and their C-ABI
// entry points. gfx950 only. The fused multi-tendency RHS kernels live in k_rhs.hip.
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include "k_common.h"

namespace mhh
{
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
}
using namespace mhh;

MHH_API const char* mhh_last_error(void) { return mhh::g_err; }
MHH_API int mhh_version(void) { return 100; }
MHH_API int mhh_synchronize(void* stream) { MHH_HIP_TRY(hipStreamSynchronize(mhh::as_stream(stream))); return MHH_OK; }

// =======================================================================================================
// Boundary_cyclic (src/boundary_cyclic.cxx:370-443; GPU counterpart src/boundary_cyclic.cu:30-127)
// One thread per ghost cell, i fastest. East-west first over ALL j,k; then north-south over all i incl.
// the x ghosts, so corners end up right. Several fields per launch via blockIdx.z / nplanes.
// =======================================================================================================
constexpr int MAXF = 8;
template<class TF> struct FieldList { TF* f[MAXF]; };

template<class TF>
__global__ void __launch_bounds__(256) cyclic_x_kernel(FieldList<TF> fl, int igc, int iend, int istart, int icells, int nrows, int tpr)
{
    // a row = one (j,k) line; each thread copies one west and one east ghost cell; tpr = threads per row (8, 16 or 32 >= igc)
    const int t = threadIdx.x % tpr;
    const int row = blockIdx.x * (256/tpr) + threadIdx.x / tpr;
    if (t >= igc || row >= nrows) return;
    TF* __restrict__ a = fl.f[blockIdx.y] + (size_t)row * icells;
    a[t] = a[iend - igc + t];
    a[iend + t] = a[istart + t];
}
template<class TF>
__global__ void __launch_bounds__(256) cyclic_y_kernel(FieldList<TF> fl, int jgc, int jend, int jstart, int icells, int ijcells, int kcells)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (i >= icells) return;
    TF* __restrict__ a = fl.f[blockIdx.z] + (size_t)k * ijcells;
    for (int j=0; j<jgc; ++j)
    {
        a[i + j*icells] = a[i + (jend - jgc + j)*icells];
        a[i + (jend + j)*icells] = a[i + (jstart + j)*icells];
    }
}
// jtot == 1: replicate the single row over the y ghosts (interior k only, as the reference)
template<class TF>
__global__ void __launch_bounds__(256) cyclic_y2d_kernel(FieldList<TF> fl, int jgc, int jend, int jstart, int icells, int ijcells, int kstart)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int k = kstart + blockIdx.y;
    if (i >= icells) return;
    TF* __restrict__ a = fl.f[blockIdx.z] + (size_t)k * ijcells;
    const TF r = a[i + jstart*icells];
    for (int j=0; j<jgc; ++j) { a[i + j*icells] = r; a[i + (jend + j)*icells] = r; }
}

template<class TF>
static int cyclic_launch(const mhh_grid* g, void* const* data, int nf, int edge, int kcells, int kstart, int kend, hipStream_t st)
{
    MHH_REQUIRE(nf >= 1 && nf <= MAXF, "1..8 fields per call");
    MHH_REQUIRE(g->igc <= 32 && g->jgc <= 32, "ghost width <= 32");
    const int tpr = (g->igc <= 8) ? 8 : (g->igc <= 16 ? 16 : 32), rpb = 256 / tpr;
    MHH_REQUIRE(g->imax >= g->igc && (g->jtot == 1 || g->jmax >= g->jgc), "the interior must be at least as wide as the ghost zone it fills (src/grid.cxx:420)");
    FieldList<TF> fl;
    for (int n=0; n<MAXF; ++n) fl.f[n] = mp<TF>(data[n < nf ? n : 0]);
    for (int n=0; n<nf; ++n) MHH_REQUIRE(data[n] != nullptr, "null field");
    if (edge == MHH_EDGE_EW || edge == MHH_EDGE_BOTH)
    {
        const int nrows = g->jcells * kcells;
        hipLaunchKernelGGL(cyclic_x_kernel<TF>, dim3((nrows + rpb-1)/rpb, nf), dim3(256), 0, st, fl, g->igc, g->iend, g->istart, g->icells, nrows, tpr);
        MHH_LAUNCH_CHECK();
    }
    if (edge == MHH_EDGE_NS || edge == MHH_EDGE_BOTH)
    {
        if (g->jtot > 1)
            hipLaunchKernelGGL(cyclic_y_kernel<TF>, dim3((g->icells + 255)/256, kcells, nf), dim3(256), 0, st, fl, g->jgc, g->jend, g->jstart, g->icells, g->ijcells, kcells);
        else if (kend > kstart)
            hipLaunchKernelGGL(cyclic_y2d_kernel<TF>, dim3((g->icells + 255)/256, kend-kstart, nf), dim3(256), 0, st, fl, g->jgc, g->jend, g->jstart, g->icells, g->ijcells, kstart);
        MHH_LAUNCH_CHECK();
    }
    return MHH_OK;
}

MHH_API int mhh_boundary_cyclic_n(const mhh_grid* g, void* const* data, int nfields, int edge, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(edge >= 0 && edge <= 2, "edge");
    MHH_REQUIRE(g->npy == 1 || edge == MHH_EDGE_EW, "slab-decomposed grid: north-south ghosts come from the neighbour exchange (mhh_halo_pack_ns / mhh_halo_unpack_ns)");
    if (g->dtype == MHH_F64) return cyclic_launch<double>(g, data, nfields, edge, g->kcells, g->kstart, g->kend, as_stream(stream));
    return cyclic_launch<float>(g, data, nfields, edge, g->kcells, g->kstart, g->kend, as_stream(stream));
}
MHH_API int mhh_boundary_cyclic(const mhh_grid* g, void* data, int edge, void* stream)
{
    void* d[1] = {data};
    return mhh_boundary_cyclic_n(g, d, 1, edge, stream);
}
MHH_API int mhh_boundary_cyclic_2d(const mhh_grid* g, void* data, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(g->npy == 1, "slab-decomposed grid: the north-south ghost rows of a 2-D field come from the neighbour exchange (mhh_halo_pack_rows / mhh_halo_unpack_rows), not from a local wrap");
    void* d[1] = {data};
    // one slice: kcells = 1 and the jtot == 1 branch runs on that slice
    if (g->dtype == MHH_F64) return cyclic_launch<double>(g, d, 1, MHH_EDGE_BOTH, 1, 0, 1, as_stream(stream));
    return cyclic_launch<float>(g, d, 1, MHH_EDGE_BOTH, 1, 0, 1, as_stream(stream));
}

// the unsigned-int variants: a ghost-cell fill only copies, so a 32-bit integer field is filled as 4-byte values, bit for bit
// (loads and stores of float registers do not touch the bits, signalling-NaN patterns included)
MHH_API int mhh_boundary_cyclic_u32(const mhh_grid* g, void* data, int edge, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(edge >= 0 && edge <= 2, "edge");
    MHH_REQUIRE(g->npy == 1 || edge == MHH_EDGE_EW, "slab-decomposed grid: north-south ghosts come from the neighbour exchange");
    void* d[1] = {data};
    return cyclic_launch<float>(g, d, 1, edge, g->kcells, g->kstart, g->kend, as_stream(stream));
}
MHH_API int mhh_boundary_cyclic_2d_u32(const mhh_grid* g, void* data, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(g->npy == 1, "slab-decomposed grid: the north-south ghost rows of a 2-D field come from the neighbour exchange");
    void* d[1] = {data};
    return cyclic_launch<float>(g, d, 1, MHH_EDGE_BOTH, 1, 0, 1, as_stream(stream));
}

// =======================================================================================================
// Advection, one tendency per launch
// =======================================================================================================
template<class TF>
struct AdvecOp
{
    GridDev<TF> g; int scheme; int comp;           // comp 0..2 momentum, 3 scalar
    TF* __restrict__ t; const TF* __restrict__ f;
    const TF* __restrict__ u; const TF* __restrict__ v; const TF* __restrict__ w;
    const TF* __restrict__ rhoref; const TF* __restrict__ rhorefh;

    __device__ void operator()(int, int, int k, int c) const
    {
        const int jj = g.icells, kk = g.ijcells;
        const bool isw = (comp == 2);
        if (scheme == MHH_ADVEC_4M)
        {
            const bool bot = (k == g.kstart), top = (k == g.kend-1);
            t[c] += advec4m(comp, f, u, v, w, c, jj, kk, bot, top, g.dxi_d, g.dyi_d, isw ? g.dzhi4[k] : g.dzi4[k]);
            return;
        }
        if (scheme == MHH_ADVEC_4)
        {
            const int k0 = isw ? g.kstart+1 : g.kstart;
            const bool bot = (k == k0), top = (k == g.kend-1);
            const TF dz = isw ? g.dzhi4[k] : g.dzi4[k];
            TF d[3];
            if (comp == 3) advec4_s(d, f, u, v, w, c, jj, kk, bot, top, g.dxi_t, g.dyi_t, dz, g.dim3);
            else           advec4_mom(d, f, u, v, w, c, comp==0 ? 1 : (comp==1 ? jj : kk), isw, jj, kk, bot, top, g.dxi_t, g.dyi_t, dz, g.dim3);
            TF x = t[c];
            x -= d[0];
            if (g.dim3) x -= d[1];
            x -= d[2];
            t[c] = x;
            return;
        }
        const TF rt = isw ? rhoref[k]   : rhorefh[k+1];
        const TF rb = isw ? rhoref[k-1] : rhorefh[k];
        const TF rc = isw ? rhorefh[k]  : rhoref[k];
        const TF dz = isw ? g.dzhi[k]   : g.dzi[k];
        const int o = comp==0 ? -1 : (comp==1 ? -jj : -kk);
        if (scheme == MHH_ADVEC_2)
        {
            if (comp == 3) t[c] += advec2_s(f, u, v, w, c, jj, kk, g.dxi_t, g.dyi_t, rt, rb, rc, dz);
            else           t[c] += advec2_mom(f, u, v, w, c, o, jj, kk, g.dxi_t, g.dyi_t, rt, rb, rc, dz);
            return;
        }
        // 2i5: horizontal then vertical, two separate accumulations (src/advec_2i5.cxx:187,210); 2i4: one accumulation
        TF ue, uw, vn, vs, wtp, wbt; int ot, ob;
        if (comp == 3)
        {
            ue = u[c+1]; uw = u[c]; vn = v[c+jj]; vs = v[c]; wtp = w[c+kk]; wbt = w[c];
            ot = order_face_c(k+1, g.kstart, g.kend); ob = order_face_c(k, g.kstart, g.kend);
        }
        else
        {
            ue = i2(u[c+1+o], u[c+1]);   uw = i2(u[c+o], u[c]);
            vn = i2(v[c+jj+o], v[c+jj]); vs = i2(v[c+o], v[c]);
            wtp = i2(w[c+kk+o], w[c+kk]); wbt = i2(w[c+o], w[c]);
            if (isw) { ot = order_face_w(k, g.kstart, g.kend); ob = order_face_w(k-1, g.kstart, g.kend); }
            else     { ot = order_face_c(k+1, g.kstart, g.kend); ob = order_face_c(k, g.kstart, g.kend); }
        }
        if (scheme == MHH_ADVEC_2I53) { ot = ot > 4 ? 4 : ot; ob = ob > 4 ? 4 : ob; }     // src/advec_2i53.cxx: 4th/3rd order vertically
        if (scheme == MHH_ADVEC_2I62)
        {
            t[c] += advec262(f, c, jj, kk, ue, uw, vn, vs, wtp, wbt, g.dxi_t, g.dyi_t, rt, rb, rc, dz);
            return;
        }
        if (scheme == MHH_ADVEC_2I4)
        {
            t[c] += advec24(f, c, jj, kk, ue, uw, vn, vs, wtp, wbt, ot > 4 ? 4 : ot, ob > 4 ? 4 : ob, g.dxi_d, g.dyi_d, rt, rb, rc, dz);
            return;
        }
        TF x = t[c];
        x += advec25_hor(f, c, jj, ue, uw, vn, vs, g.dxi_t, g.dyi_t);
        x += advec25_ver(f, c, kk, ot, ob, wtp, wbt, rt, rb, rc, dz);
        t[c] = x;
    }
};

static int check_advec(const mhh_grid* g, int scheme)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(scheme == MHH_ADVEC_2 || scheme == MHH_ADVEC_2I5 || scheme == MHH_ADVEC_4 || scheme == MHH_ADVEC_2I4 || scheme == MHH_ADVEC_2I62 || scheme == MHH_ADVEC_2I53 || scheme == MHH_ADVEC_4M, "scheme must be 2, 24, 25, 253, 262, 4 or 41");
    if (scheme == MHH_ADVEC_4M) MHH_REQUIRE(g->igc >= 3 && g->jgc >= 3 && g->kgc >= 3 && g->ktot >= 2, "advec_4m needs gc(3,3,3)");
    if (scheme == MHH_ADVEC_2I53) MHH_REQUIRE(g->igc >= 3 && g->jgc >= 3 && g->kgc >= 1 && g->ktot >= 4, "advec_2i53 needs gc(3,3,1), ktot >= 4");
    if (scheme == MHH_ADVEC_2I62) MHH_REQUIRE(g->igc >= 3 && g->jgc >= 3 && g->kgc >= 1, "advec_2i62 needs gc(3,3,1) (src/advec_2i62.cxx:42-45)");
    if (scheme == MHH_ADVEC_2I4) MHH_REQUIRE(g->igc >= 2 && g->jgc >= 2 && g->kgc >= 1 && g->ktot >= 4, "advec_2i4 needs gc(2,2,1) and ktot >= 4 (the reference asks for gc(2,2,2), src/advec_2i4.cxx:38-41)");
    if (scheme == MHH_ADVEC_2)   MHH_REQUIRE(g->igc >= 1 && g->jgc >= 1 && g->kgc >= 1, "advec_2 needs 1 ghost cell");
    if (scheme == MHH_ADVEC_2I5) MHH_REQUIRE(g->igc >= 3 && g->jgc >= 3 && g->kgc >= 1 && g->ktot >= 6, "advec_2i5 needs gc(3,3,1), ktot>=6 (src/advec_2i5.cxx:42-45)");
    if (scheme == MHH_ADVEC_4)   MHH_REQUIRE(g->igc >= 3 && g->jgc >= 3 && g->kgc >= 3 && g->ktot >= 2, "advec_4 needs gc(3,3,3)");
    return MHH_OK;
}

template<class TF>
static int advec_launch(const mhh_grid* g, int scheme, int comp, void* t, const void* f, const void* u, const void* v, const void* w,
                        const void* r, const void* rh, void* stream)
{
    AdvecOp<TF> op{make_grid<TF>(g), scheme, comp, mp<TF>(t), cp<TF>(f), cp<TF>(u), cp<TF>(v), cp<TF>(w), cp<TF>(r), cp<TF>(rh)};
    return launch_interior(as_stream(stream), op.g, comp == 2 ? g->kstart+1 : g->kstart, g->kend, op);
}
static int advec_any(const mhh_grid* g, int scheme, int comp, void* t, const void* f, const void* u, const void* v, const void* w,
                     const void* r, const void* rh, void* stream)
{
    if (int e = check_advec(g, scheme)) return e;
    MHH_REQUIRE(t && f && u && v && w, "null field");
    MHH_REQUIRE(scheme == MHH_ADVEC_4 || scheme == MHH_ADVEC_4M || (r && rh), "rhoref/rhorefh required");
    if (g->dtype == MHH_F64) return advec_launch<double>(g, scheme, comp, t, f, u, v, w, r, rh, stream);
    return advec_launch<float>(g, scheme, comp, t, f, u, v, w, r, rh, stream);
}
MHH_API int mhh_advec_u(const mhh_grid* g, int scheme, void* ut, const void* u, const void* v, const void* w, const void* r, const void* rh, void* s)
{ return advec_any(g, scheme, 0, ut, u, u, v, w, r, rh, s); }
MHH_API int mhh_advec_v(const mhh_grid* g, int scheme, void* vt, const void* u, const void* v, const void* w, const void* r, const void* rh, void* s)
{ return advec_any(g, scheme, 1, vt, v, u, v, w, r, rh, s); }
MHH_API int mhh_advec_w(const mhh_grid* g, int scheme, void* wt, const void* u, const void* v, const void* w, const void* r, const void* rh, void* s)
{ return advec_any(g, scheme, 2, wt, w, u, v, w, r, rh, s); }
MHH_API int mhh_advec_s(const mhh_grid* g, int scheme, void* st, const void* sc, const void* u, const void* v, const void* w, const void* r, const void* rh, void* s)
{ return advec_any(g, scheme, 3, st, sc, u, v, w, r, rh, s); }

// flux-limited scalar advection (include/advec_monotonic.h:79-180)
template<class TF>
struct AdvecSLimOp
{
    GridDev<TF> g; TF* __restrict__ t; const TF* __restrict__ s;
    const TF* __restrict__ u; const TF* __restrict__ v; const TF* __restrict__ w;
    const TF* __restrict__ rhoref; const TF* __restrict__ rhorefh;
    __device__ void operator()(int, int, int k, int c) const
    {
        const int lev = (k == g.kstart) ? 1 : (k == g.kstart+1) ? 2 : (k == g.kend-2) ? 3 : (k == g.kend-1) ? 4 : 0;
        t[c] += advec_s_lim_cell(s, u, v, w, c, g.icells, g.ijcells, lev, g.dxi_t, g.dyi_t, rhorefh[k+1], rhorefh[k], rhoref[k], g.dzi[k]);
    }
};
MHH_API int mhh_advec_s_lim(const mhh_grid* g, void* st, const void* sc, const void* u, const void* v, const void* w, const void* r, const void* rh, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(st && sc && u && v && w && r && rh, "null field");
    MHH_REQUIRE(g->igc >= 2 && g->jgc >= 2 && g->kgc >= 1 && g->ktot >= 4, "advec_s_lim needs 2 horizontal ghost cells, 1 vertical, ktot >= 4");
#define CALL(TF) [&]{ AdvecSLimOp<TF> op{make_grid<TF>(g), mp<TF>(st), cp<TF>(sc), cp<TF>(u), cp<TF>(v), cp<TF>(w), cp<TF>(r), cp<TF>(rh)}; \
        return launch_interior(as_stream(stream), op.g, g->kstart, g->kend, op); }()
    return MHH_DISPATCH(g, CALL);
#undef CALL
}

int mhh_advec25_march(const mhh_grid* g, const mhh_fields* f, void* stream);                    // k_march.hip
int mhh_advec4_march(const mhh_grid* g, const mhh_fields* f, void* stream);                     // k_march4.hip
MHH_API int mhh_advec_exec(const mhh_grid* g, int scheme, const mhh_fields* f, void* stream)
{
    MHH_REQUIRE(f != nullptr, "fields");
    MHH_REQUIRE(f->nscalars >= 0 && f->nscalars <= MHH_MAX_SCALARS, "nscalars");
    // advec_2i5: u, v, w and the first (unlimited) scalar in one pass of the marching kernel with the advective terms only
    // (k_march.hip; same bits as the per-field kernels, which MHH_ADVEC25_IMPL=cell selects); further scalars per field.
    const char* impl = getenv("MHH_ADVEC25_IMPL");
    if (scheme == MHH_ADVEC_2I5 && !(impl && !strcmp(impl, "cell")))
    {
        if (int e = check_grid(g)) return e;
        if (int e = check_advec(g, scheme)) return e;
        MHH_REQUIRE(f->u && f->v && f->w && f->ut && f->vt && f->wt && f->rhoref && f->rhorefh, "null field");
        for (int n=0; n<f->nscalars; ++n) MHH_REQUIRE(f->s[n] && f->st[n], "null scalar");
        mhh_fields fm = *f;
        const bool s0 = f->nscalars > 0 && !f->s_fluxlimit[0];
        fm.nscalars = s0 ? 1 : 0;
        if (int e = mhh_advec25_march(g, &fm, stream)) return e;
        for (int n=(s0 ? 1 : 0); n<f->nscalars; ++n)
        {
            if (f->s_fluxlimit[n]) { if (int e = mhh_advec_s_lim(g, f->st[n], f->s[n], f->u, f->v, f->w, f->rhoref, f->rhorefh, stream)) return e; }
            else if (int e = mhh_advec_s(g, scheme, f->st[n], f->s[n], f->u, f->v, f->w, f->rhoref, f->rhorefh, stream)) return e;
        }
        return MHH_OK;
    }
    bool uvw_done = false;
    if (scheme == MHH_ADVEC_4)           // u, v, w in one pass of the 4th-order marching kernel (advective terms only); scalars per field
    {
        if (int e = check_advec(g, scheme)) return e;
        MHH_REQUIRE(f->u && f->v && f->w && f->ut && f->vt && f->wt, "null field");
        const int rc = mhh_advec4_march(g, f, stream);
        if (rc < 0) return -rc;
        uvw_done = (rc == 1);
    }
    if (!uvw_done)
    {
        if (int e = mhh_advec_u(g, scheme, f->ut, f->u, f->v, f->w, f->rhoref, f->rhorefh, stream)) return e;
        if (int e = mhh_advec_v(g, scheme, f->vt, f->u, f->v, f->w, f->rhoref, f->rhorefh, stream)) return e;
        if (int e = mhh_advec_w(g, scheme, f->wt, f->u, f->v, f->w, f->rhoref, f->rhorefh, stream)) return e;
    }
    for (int n=0; n<f->nscalars; ++n)
    {
        if (f->s_fluxlimit[n])
        {
            MHH_REQUIRE(scheme == MHH_ADVEC_2I5 || scheme == MHH_ADVEC_2I62, "fluxlimit_list is an option of advec_2i5 / advec_2i62 (src/advec_2i5.cxx:39, src/advec_2i62.cxx:39)");
            if (int e = mhh_advec_s_lim(g, f->st[n], f->s[n], f->u, f->v, f->w, f->rhoref, f->rhorefh, stream)) return e;
        }
        else if (int e = mhh_advec_s(g, scheme, f->st[n], f->s[n], f->u, f->v, f->w, f->rhoref, f->rhorefh, stream)) return e;
    }
    return MHH_OK;
}

// =======================================================================================================
// diff_2 / diff_4
// =======================================================================================================
template<class TF>
struct DiffOp
{
    GridDev<TF> g; int order; int is_w; TF visc;
    TF* __restrict__ t; const TF* __restrict__ a;
    __device__ void operator()(int, int, int k, int c) const
    {
        const int jj = g.icells, kk = g.ijcells;
        if (order == 2)
        {
            const TF gt = is_w ? g.dzi[k]   : g.dzhi[k+1];
            const TF gb = is_w ? g.dzi[k-1] : g.dzhi[k];
            const TF gc = is_w ? g.dzhi[k]  : g.dzi[k];
            t[c] = diff2_apply(t[c], a, c, jj, kk, visc, g.dxidxi_2, g.dyidyi_2, gt, gb, gc);
            return;
        }
        const int k0 = is_w ? g.kstart+1 : g.kstart;
        const bool bot = (k == k0), top = (k == g.kend-1);
        const TF* gi = is_w ? g.dzi4 : g.dzhi4;
        const int s = is_w ? -1 : 0;
        const TF g4[4] = {gi[k-1+s], gi[k+s], gi[k+1+s], gi[k+2+s]};
        const TF go = is_w ? g.dzhi4[k] : g.dzi4[k];
        TF d[3];
        diff4_cell(d, a, c, jj, kk, bot, top, visc, is_w ? g.dxidxi_t : g.dxidxi_d, is_w ? g.dyidyi_t : g.dyidyi_d, g4, go, g.dim3);
        TF x = t[c];
        x += d[0];
        if (g.dim3) x += d[1];
        x += d[2];
        t[c] = x;
    }
};
template<class TF>
static int diff_launch(const mhh_grid* g, int order, int is_w, void* t, const void* a, double visc, void* stream)
{
    DiffOp<TF> op{make_grid<TF>(g), order, is_w, TF(visc), mp<TF>(t), cp<TF>(a)};
    return launch_interior(as_stream(stream), op.g, is_w ? g->kstart+1 : g->kstart, g->kend, op);
}
static int diff_any(const mhh_grid* g, int order, int is_w, void* t, const void* a, double visc, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(order == 2 || order == 4, "order must be 2 or 4");
    MHH_REQUIRE(t && a, "null field");
    if (order == 2) MHH_REQUIRE(g->igc >= 1 && g->jgc >= 1 && g->kgc >= 1, "diff_2 needs 1 ghost cell");
    else            MHH_REQUIRE(g->igc >= 3 && g->jgc >= 3 && g->kgc >= 3, "diff_4 needs gc(3,3,3)");
    if (g->dtype == MHH_F64) return diff_launch<double>(g, order, is_w, t, a, visc, stream);
    return diff_launch<float>(g, order, is_w, t, a, visc, stream);
}
MHH_API int mhh_diff_c(const mhh_grid* g, int order, void* at, const void* a, double visc, void* stream) { return diff_any(g, order, 0, at, a, visc, stream); }
MHH_API int mhh_diff_w(const mhh_grid* g, int order, void* wt, const void* w, double visc, void* stream) { return diff_any(g, order, 1, wt, w, visc, stream); }

// =======================================================================================================
// diff_smag2 pieces
// =======================================================================================================
template<class TF>
struct Strain2Op
{
    GridDev<TF> g; int sm; TF* __restrict__ s2;
    const TF* __restrict__ u; const TF* __restrict__ v; const TF* __restrict__ w;
    const TF* __restrict__ dudz; const TF* __restrict__ dvdz;
    __device__ void operator()(int i, int j, int k, int c) const
    {
        const bool mo = sm && (k == g.kstart);
        const int ij = i + j*g.icells;
        s2[c] = smag_strain2(u, v, w, c, g.icells, g.ijcells, mo, mo ? dudz[ij] : TF(0), mo ? dvdz[ij] : TF(0),
                             g.dxi_d, g.dyi_d, g.dzi[k], g.dzhi[k], g.dzhi[k+1]);
    }
};
static int check_smag(const mhh_grid* g)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(g->igc >= 1 && g->jgc >= 1 && g->kgc >= 1, "diff_smag2 needs 1 ghost cell");
    return MHH_OK;
}
MHH_API int mhh_smag2_strain2(const mhh_grid* g, int sm, void* s2, const void* u, const void* v, const void* w, const void* dudz, const void* dvdz, void* stream)
{
    if (int e = check_smag(g)) return e;
    MHH_REQUIRE(s2 && u && v && w, "null field");
    MHH_REQUIRE(!sm || (dudz && dvdz), "surface model needs dudz, dvdz");
#define CALL(TF) [&]{ Strain2Op<TF> op{make_grid<TF>(g), sm, mp<TF>(s2), cp<TF>(u), cp<TF>(v), cp<TF>(w), cp<TF>(dudz), cp<TF>(dvdz)}; \
                      return launch_interior(as_stream(stream), op.g, g->kstart, g->kend, op); }()
    return MHH_DISPATCH(g, CALL);
#undef CALL
}

// calc_evisc (src/diff_smag2.cxx:254-367). mlen0 = cs*pow(dx*dy*dz[k],1/3) per level comes from the host
// (Diff_smag2::prepare_device computes the same table on the host, src/diff_smag2.cu:521-542) so that the
// libm call is the CPU's; Mason's n = 2 lets pow(x,2) / pow(y,1/2) be an exact square and a correctly
// rounded sqrt (see DESIGN.md "Parity").
template<class TF> __device__ __forceinline__ TF dsqrt(TF x);
template<> __device__ __forceinline__ double dsqrt<double>(double x) { return __builtin_sqrt(x); }
template<> __device__ __forceinline__ float  dsqrt<float>(float x)   { return __builtin_sqrtf(x); }

template<class TF>
struct EviscOp
{
    GridDev<TF> g; int sm; TF* __restrict__ ev;
    const TF* __restrict__ N2; const TF* __restrict__ bgradbot; const TF* __restrict__ z0m; const TF* __restrict__ mlen0;
    TF tPr;
    __device__ void operator()(int i, int j, int k, int c) const
    {
        const int ij = i + j*g.icells;
        const TF s2 = ev[c];
        TF rit = ((sm && k == g.kstart) ? bgradbot[ij] : N2[c]) / s2 / tPr;
        rit = tmin(rit, TF(1.-1.e-9));
        TF fac;
        if (!sm) fac = sq(mlen0[k]);
        else
        {
            const TF kz = TF(0.4)*(g.z[k]+z0m[ij]);
            const TF mlen = dsqrt(TF(1.)/(TF(1.)/sq(mlen0[k]) + TF(1.)/sq(kz)));
            fac = sq(mlen);
        }
        ev[c] = fac * dsqrt(s2) * dsqrt(TF(1.)-rit);
    }
};
template<class TF>
struct EviscNeutralOp     // calc_evisc_neutral with surface model (:229-249); Mason n = 1
{
    GridDev<TF> g; TF* __restrict__ ev; const TF* __restrict__ z0m; const TF* __restrict__ mlen0;
    __device__ void operator()(int i, int j, int k, int c) const
    {
        const int ij = i + j*g.icells;
        const TF mlen = TF(1.)/(TF(1.)/mlen0[k] + TF(1.)/(TF(0.4)*(g.z[k]+z0m[ij])));
        ev[c] = sq(mlen) * dsqrt(ev[c]);
    }
};
template<class TF> __device__ __forceinline__ TF dpow(TF x, TF y);
template<> __device__ __forceinline__ double dpow<double>(double x, double y) { return pow(x, y); }
template<> __device__ __forceinline__ float  dpow<float>(float x, float y)    { return powf(x, y); }
template<class TF> __device__ __forceinline__ TF dexp(TF x);
template<> __device__ __forceinline__ double dexp<double>(double x) { return exp(x); }
template<> __device__ __forceinline__ float  dexp<float>(float x)   { return expf(x); }

template<class TF>
struct EviscNeutralWallOp  // calc_evisc_neutral, resolved walls with van Driest damping (:181-212)
{
    GridDev<TF> g; TF* __restrict__ ev; const TF* __restrict__ u; const TF* __restrict__ v; const TF* __restrict__ mlen0; TF visc;
    __device__ void operator()(int i, int j, int k, int c) const
    {
        const int jj = g.icells, kk = g.ijcells;
        const int cb = i + j*jj + g.kstart*kk, ct = i + j*jj + g.kend*kk;
        const TF A = TF(26.);
        const TF utb = dpow( sq( visc*(u[cb] - u[cb-kk])*g.dzhi[g.kstart] ) + sq( visc*(v[cb] - v[cb-kk])*g.dzhi[g.kstart] ), TF(0.25) );
        const TF utt = dpow( sq( visc*(u[ct] - u[ct-kk])*g.dzhi[g.kend] ) + sq( visc*(v[ct] - v[ct-kk])*g.dzhi[g.kend] ), TF(0.25) );
        const TF fb = TF(1.) - dexp( -(          g.z[k] *utb) / (A*visc) );
        const TF ft = TF(1.) - dexp( -((g.zsize-g.z[k])*utt) / (A*visc) );
        const TF fac = tmin(fb, ft);
        ev[c] = sq(fac * mlen0[k]) * dsqrt(ev[c]);
    }
};
template<class TF>
struct MirrorWallOp        // evisc ghost levels in resolved-wall mode (:218-226, :303-311); runs over all icells x jcells
{
    GridDev<TF> g; TF* __restrict__ ev;
    __device__ void operator()(int i, int j, int, int) const
    {
        const int b = i + j*g.icells + g.kstart*g.ijcells, t = i + j*g.icells + (g.kend-1)*g.ijcells;
        ev[b-g.ijcells] = ev[b];
        ev[t+g.ijcells] = ev[t];
    }
};

// host helper: the per-level Smagorinsky length cs*(dx*dy*dz)^(1/3), evaluated with the C library's pow like
// the reference CPU path (src/diff_smag2.cxx:273,325) -- `g` carries HOST metric pointers here.
MHH_API int mhh_smag2_mlen0_host(const mhh_grid* g, double cs, void* out)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(g->dz && out, "null pointer");
    if (g->dtype == MHH_F64)
    {
        const double dx = g->dx, dy = g->dy; const double* dz = cp<double>(g->dz); double* o = mp<double>(out);
        for (int k=0; k<g->kcells; ++k) o[k] = cs*std::pow(dx*dy*dz[k], double(1./3.));
    }
    else
    {
        const float dx = (float)g->dx, dy = (float)g->dy, c = (float)cs; const float* dz = cp<float>(g->dz); float* o = mp<float>(out);
        for (int k=0; k<g->kcells; ++k) o[k] = c*std::pow(dx*dy*dz[k], float(1./3.));
    }
    return MHH_OK;
}

// host helper: the squared mixing length per level for a horizontally uniform z0m -- evisc_mlen2 (cell_ops.h), the very
// function the kernels evaluate per cell, on IEEE-correct operations (same bits on the host as on the device).
MHH_API int mhh_smag2_mlen2_host(const mhh_grid* g, int sm, int neutral, const void* mlen0, double z0m, void* out)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(mlen0 && out && (!sm || g->z), "null pointer");
    if (g->dtype == MHH_F64)
        for (int k=0; k<g->kcells; ++k) mp<double>(out)[k] = evisc_mlen2<double>(sm, neutral, cp<double>(mlen0)[k], sm ? cp<double>(g->z)[k] : 0., z0m);
    else
        for (int k=0; k<g->kcells; ++k) mp<float>(out)[k] = evisc_mlen2<float>(sm, neutral, cp<float>(mlen0)[k], sm ? cp<float>(g->z)[k] : 0.f, (float)z0m);
    return MHH_OK;
}

template<class TF>
static int evisc_finish(const mhh_grid* g, const GridDev<TF>& gd, int sm, void* ev, hipStream_t st)
{
    if (!sm)
    {
        MirrorWallOp<TF> m{gd, mp<TF>(ev)};
        if (int e = launch_cells(st, m, 0, g->icells, 0, g->jcells, 0, 1, g->icells, g->ijcells)) return e;
    }
    // slab-decomposed: only the local east-west wrap; the caller exchanges the north-south halo
    return mhh_boundary_cyclic(g, ev, g->npy > 1 ? MHH_EDGE_EW : MHH_EDGE_BOTH, st);
}
MHH_API int mhh_smag2_evisc(const mhh_grid* g, int sm, void* ev, const void* N2, const void* bgradbot, const void* z0m,
                            const void* mlen0, double tPr, void* stream)
{
    if (int e = check_smag(g)) return e;
    MHH_REQUIRE(ev && N2 && mlen0, "null field");
    MHH_REQUIRE(!sm || (bgradbot && z0m), "surface model needs bgradbot, z0m");
#define CALL(TF) [&]{ EviscOp<TF> op{make_grid<TF>(g), sm, mp<TF>(ev), cp<TF>(N2), cp<TF>(bgradbot), cp<TF>(z0m), cp<TF>(mlen0), TF(tPr)}; \
                      if (int e = launch_interior(as_stream(stream), op.g, g->kstart, g->kend, op)) return e; \
                      return evisc_finish<TF>(g, op.g, sm, ev, as_stream(stream)); }()
    return MHH_DISPATCH(g, CALL);
#undef CALL
}
MHH_API int mhh_smag2_evisc_neutral(const mhh_grid* g, int sm, void* ev, const void* u, const void* v, const void* z0m,
                                    const void* mlen0, double visc, void* stream)
{
    if (int e = check_smag(g)) return e;
    MHH_REQUIRE(ev && mlen0, "null field");
    MHH_REQUIRE(sm ? (z0m != nullptr) : (u && v), "inputs");
#define CALL(TF) [&]{ GridDev<TF> gd = make_grid<TF>(g); int e; \
                      if (sm) { EviscNeutralOp<TF> op{gd, mp<TF>(ev), cp<TF>(z0m), cp<TF>(mlen0)}; e = launch_interior(as_stream(stream), gd, g->kstart, g->kend, op); } \
                      else    { EviscNeutralWallOp<TF> op{gd, mp<TF>(ev), cp<TF>(u), cp<TF>(v), cp<TF>(mlen0), TF(visc)}; e = launch_interior(as_stream(stream), gd, g->kstart, g->kend, op); } \
                      if (e) return e; return evisc_finish<TF>(g, gd, sm, ev, as_stream(stream)); }()
    return MHH_DISPATCH(g, CALL);
#undef CALL
}

template<class TF>
struct SmagDiffOp
{
    GridDev<TF> g; int comp; int sm;        // comp 0 u, 1 v, 2 w, 3 scalar
    TF* __restrict__ t; const TF* __restrict__ a;
    const TF* __restrict__ u; const TF* __restrict__ v; const TF* __restrict__ w; const TF* __restrict__ ev;
    const TF* __restrict__ fluxbot; const TF* __restrict__ fluxtop;
    const TF* __restrict__ rhoref; const TF* __restrict__ rhorefh; TF visc, tPr;
    __device__ void operator()(int i, int j, int k, int c) const
    {
        const int jj = g.icells, kk = g.ijcells;
        if (comp == 2)
        {
            t[c] += smag_diff_w(u, v, w, ev, c, jj, kk, visc, g.dxi_d, g.dyi_d, rhoref[k], rhoref[k-1], rhorefh[k], g.dzi[k], g.dzi[k-1], g.dzhi[k]);
            return;
        }
        const bool fb = sm && (k == g.kstart), ft = sm && (k == g.kend-1);
        const int ij = i + j*jj;
        const TF flb = fb ? fluxbot[ij] : TF(0), flt = ft ? fluxtop[ij] : TF(0);
        if (comp == 0)      t[c] += smag_diff_u(u, v, w, ev, c, jj, kk, fb, ft, flb, flt, visc, g.dxi_d, g.dyi_d, rhorefh[k], rhorefh[k+1], rhoref[k], g.dzi[k], g.dzhi[k], g.dzhi[k+1]);
        else if (comp == 1) t[c] += smag_diff_v(u, v, w, ev, c, jj, kk, fb, ft, flb, flt, visc, g.dxi_d, g.dyi_d, rhorefh[k], rhorefh[k+1], rhoref[k], g.dzi[k], g.dzhi[k], g.dzhi[k+1]);
        else                t[c] += smag_diff_c(a, ev, c, jj, kk, fb, ft, flb, flt, tPr, visc, g.dxidxi_d, g.dyidyi_d, rhorefh[k], rhorefh[k+1], rhoref[k], g.dzi[k], g.dzhi[k], g.dzhi[k+1]);
    }
};
static int smag_diff_any(const mhh_grid* g, int comp, int sm, void* t, const void* a, const void* u, const void* v, const void* w, const void* ev,
                         const void* fb, const void* ft, const void* r, const void* rh, double visc, double tPr, void* stream)
{
    if (int e = check_smag(g)) return e;
    MHH_REQUIRE(t && ev && r && rh, "null field");
    MHH_REQUIRE(comp == 3 ? (a != nullptr) : (u && v && w), "null field");
    MHH_REQUIRE(comp == 2 || !sm || (fb && ft), "surface model needs fluxbot, fluxtop");
#define CALL(TF) [&]{ SmagDiffOp<TF> op{make_grid<TF>(g), comp, sm, mp<TF>(t), cp<TF>(a), cp<TF>(u), cp<TF>(v), cp<TF>(w), cp<TF>(ev), cp<TF>(fb), cp<TF>(ft), cp<TF>(r), cp<TF>(rh), TF(visc), TF(tPr)}; \
                      return launch_interior(as_stream(stream), op.g, comp == 2 ? g->kstart+1 : g->kstart, g->kend, op); }()
    return MHH_DISPATCH(g, CALL);
#undef CALL
}
MHH_API int mhh_smag2_diff_u(const mhh_grid* g, int sm, void* ut, const void* u, const void* v, const void* w, const void* ev, const void* fb, const void* ft, const void* r, const void* rh, double visc, void* s)
{ return smag_diff_any(g, 0, sm, ut, nullptr, u, v, w, ev, fb, ft, r, rh, visc, 1., s); }
MHH_API int mhh_smag2_diff_v(const mhh_grid* g, int sm, void* vt, const void* u, const void* v, const void* w, const void* ev, const void* fb, const void* ft, const void* r, const void* rh, double visc, void* s)
{ return smag_diff_any(g, 1, sm, vt, nullptr, u, v, w, ev, fb, ft, r, rh, visc, 1., s); }
MHH_API int mhh_smag2_diff_w(const mhh_grid* g, void* wt, const void* u, const void* v, const void* w, const void* ev, const void* r, const void* rh, double visc, void* s)
{ return smag_diff_any(g, 2, 0, wt, nullptr, u, v, w, ev, nullptr, nullptr, r, rh, visc, 1., s); }
MHH_API int mhh_smag2_diff_c(const mhh_grid* g, int sm, void* at, const void* a, const void* ev, const void* fb, const void* ft, const void* r, const void* rh, double tPr, double visc, void* s)
{ return smag_diff_any(g, 3, sm, at, a, nullptr, nullptr, nullptr, ev, fb, ft, r, rh, visc, tPr, s); }

// Thermo_dry buoyancy tendency (src/thermo_dry.cxx:165-197)
template<class TF>
struct BuoyancyOp
{
    GridDev<TF> g; int order; TF* __restrict__ wt; const TF* __restrict__ th; const TF* __restrict__ threfh; TF grav;
    __device__ void operator()(int, int, int k, int c) const { wt[c] += buoyancy_tend(th, c, g.ijcells, order, grav, threfh[k]); }
};
MHH_API int mhh_thermo_dry_buoyancy_tend(const mhh_grid* g, int order, void* wt, const void* th, const void* threfh, double grav, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(wt && th && threfh, "null field");
    MHH_REQUIRE(order == 2 || order == 4, "order must be 2 or 4");
    MHH_REQUIRE(g->kgc >= (order == 4 ? 2 : 1), "vertical ghost cells");
#define CALL(TF) [&]{ BuoyancyOp<TF> op{make_grid<TF>(g), order, mp<TF>(wt), cp<TF>(th), cp<TF>(threfh), TF(grav)}; \
        return launch_interior(as_stream(stream), op.g, g->kstart+1, g->kend, op); }()
    return MHH_DISPATCH(g, CALL);
#undef CALL
}

// Thermo_dry calc_N2 (src/thermo_dry.cxx:66-78)
template<class TF>
struct N2Op
{
    GridDev<TF> g; TF* __restrict__ N2; const TF* __restrict__ th; const TF* __restrict__ thref; TF grav;
    __device__ void operator()(int, int, int k, int c) const
    { N2[c] = grav/thref[k]*TF(0.5)*(th[c+g.ijcells] - th[c-g.ijcells])*g.dzi[k]; }
};
MHH_API int mhh_calc_N2(const mhh_grid* g, void* N2, const void* th, const void* thref, double grav, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(N2 && th && thref, "null field");
#define CALL(TF) [&]{ N2Op<TF> op{make_grid<TF>(g), mp<TF>(N2), cp<TF>(th), cp<TF>(thref), TF(grav)}; \
                      return launch_interior(as_stream(stream), op.g, g->kstart, g->kend, op); }()
    return MHH_DISPATCH(g, CALL);
#undef CALL
}

// =======================================================================================================
// Timeloop rk3/rk4 substep (src/timeloop.cxx:250-334): a += cB*dt*at; at = cA_next*at (interior) or 0 (all cells)
// =======================================================================================================
template<class TF>
__global__ void __launch_bounds__(256) rk_kernel(TF* __restrict__ a, TF* __restrict__ at, TF cA, TF cB, TF dt, int reset,
                                                 int icells, int jcells, int kcells, int istart, int iend, int jstart, int jend, int kstart, int kend)
{
    const int i = blockIdx.x*256 + threadIdx.x;
    const int j = blockIdx.y, k = blockIdx.z;
    if (i >= icells) return;
    const size_t c = i + (size_t)j*icells + (size_t)k*icells*jcells;
    const bool inside = (i >= istart && i < iend && j >= jstart && j < jend && k >= kstart && k < kend);
    if (inside)
    {
        const TF t = at[c];
        a[c] = a[c] + cB*dt*t;
        at[c] = reset ? TF(0.) : cA*t;
    }
    else if (reset)
        at[c] = TF(0.);
}
MHH_API int mhh_rk_substep(const mhh_grid* g, int rkorder, int substep, double dt, void* a, void* at, void* stream)
{
    if (int e = check_grid(g)) return e;
    double cA = 0, cB = 0; bool reset = false;
    MHH_REQUIRE(rk_coefficients(rkorder, substep, cA, cB, reset) && a && at, "rkorder 3 or 4, substep in range, fields");
    dim3 grid((g->icells+255)/256, g->jcells, g->kcells);
    if (g->dtype == MHH_F64)
        hipLaunchKernelGGL(rk_kernel<double>, grid, dim3(256), 0, as_stream(stream), mp<double>(a), mp<double>(at), cA, cB, dt, reset ? 1 : 0,
                           g->icells, g->jcells, g->kcells, g->istart, g->iend, g->jstart, g->jend, g->kstart, g->kend);
    else
        hipLaunchKernelGGL(rk_kernel<float>, grid, dim3(256), 0, as_stream(stream), mp<float>(a), mp<float>(at), (float)cA, (float)cB, (float)dt, reset ? 1 : 0,
                           g->icells, g->jcells, g->kcells, g->istart, g->iend, g->jstart, g->jend, g->kstart, g->kend);
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}

// =======================================================================================================
// Vertical ghost cells: Boundary::set_ghost_cells / set_ghost_cells_w
// (src/boundary.cxx:686-907, 919-1007; GPU src/boundary.cu:119-300). bc: 0 Dirichlet, 1 Neumann / flux.
// =======================================================================================================
template<class TF>
struct GhostOp
{
    GridDev<TF> g; int order, bcbot, bctop; TF* __restrict__ a;
    const TF* __restrict__ abot; const TF* __restrict__ agradbot; const TF* __restrict__ atop; const TF* __restrict__ agradtop;
    const TF* __restrict__ dzh;
    __device__ void operator()(int i, int j, int, int) const
    {
        const int jj = g.icells, kk = g.ijcells, ks = g.kstart, ke = g.kend;
        const int ij = i + j*jj, b = ij + ks*kk, t = ij + (ke-1)*kk;
        if (order == 2)
        {
            if (bcbot == 0) a[b-kk] = TF(2.)*abot[ij] - a[b]; else a[b-kk] = -agradbot[ij]*dzh[ks] + a[b];
            if (bctop == 0) a[t+kk] = TF(2.)*atop[ij] - a[t]; else a[t+kk] = agradtop[ij]*dzh[ke] + a[t];
        }
        else
        {
            const TF cg0 = TF(1./24.), cg1 = TF(-27./24.);
            if (bcbot == 0) { a[b-kk] = TF(8./3.)*abot[ij] - TF(2.)*a[b] + TF(1./3.)*a[b+kk]; a[b-2*kk] = TF(8.)*abot[ij] - TF(9.)*a[b] + TF(2.)*a[b+kk]; }
            else
            {
                const TF gr = ( - cg0*(g.z[ks+1]-g.z[ks-2]) - cg1*(g.z[ks]-g.z[ks-1]) );
                a[b-kk] = TF(-1.)*gr*agradbot[ij] + a[b]; a[b-2*kk] = TF(-3.)*gr*agradbot[ij] + a[b+kk];
            }
            if (bctop == 0) { a[t+kk] = TF(8./3.)*atop[ij] - TF(2.)*a[t] + TF(1./3.)*a[t-kk]; a[t+2*kk] = TF(8.)*atop[ij] - TF(9.)*a[t] + TF(2.)*a[t-kk]; }
            else
            {
                const TF gr = ( - cg0*(g.z[ke+1]-g.z[ke-2]) - cg1*(g.z[ke]-g.z[ke-1]) );
                a[t+kk] = TF(1.)*gr*agradtop[ij] + a[t]; a[t+2*kk] = TF(3.)*gr*agradtop[ij] + a[t-kk];
            }
        }
    }
};
MHH_API int mhh_boundary_ghost_cells(const mhh_grid* g, int order, void* a, int bcbot, int bctop,
                                     const void* abot, const void* agradbot, const void* atop, const void* agradtop, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(order == 2 || order == 4, "order");
    MHH_REQUIRE(a && (bcbot == 0 ? abot : agradbot) && (bctop == 0 ? atop : agradtop), "null field");
    MHH_REQUIRE((bcbot == 0 || bcbot == 1) && (bctop == 0 || bctop == 1), "bc type: 0 Dirichlet, 1 Neumann");
    MHH_REQUIRE(g->kgc >= (order == 2 ? 1 : 2) && g->dzh, "ghost levels");
#define CALL(TF) [&]{ GhostOp<TF> op{make_grid<TF>(g), order, bcbot, bctop, mp<TF>(a), cp<TF>(abot), cp<TF>(agradbot), cp<TF>(atop), cp<TF>(agradtop), cp<TF>(g->dzh)}; \
                      return launch_cells(as_stream(stream), op, 0, g->icells, 0, g->jcells, 0, 1, g->icells, g->ijcells); }()
    return MHH_DISPATCH(g, CALL);
#undef CALL
}
template<class TF>
struct GhostWOp
{
    GridDev<TF> g; int type; TF* __restrict__ w;
    __device__ void operator()(int i, int j, int, int) const
    {
        const int kk = g.ijcells, b = i + j*g.icells + g.kstart*kk, t = i + j*g.icells + g.kend*kk;
        if (type == 1) { w[b-kk] = -w[b+kk]; w[b-2*kk] = -w[b+2*kk]; w[t+kk] = -w[t-kk]; w[t+2*kk] = -w[t-2*kk]; }
        else { w[b-kk] = TF(-6.)*w[b+kk] + TF(4.)*w[b+2*kk] - w[b+3*kk]; w[t+kk] = TF(-6.)*w[t-kk] + TF(4.)*w[t-2*kk] - w[t-3*kk]; }
    }
};
MHH_API int mhh_boundary_ghost_cells_w(const mhh_grid* g, void* w, int type, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(w && (type == 0 || type == 1) && g->kgc >= 2 && g->kmax >= 3, "4th-order w ghost cells: type 0 Normal / 1 Conservation, kgc >= 2");
#define CALL(TF) [&]{ GhostWOp<TF> op{make_grid<TF>(g), type, mp<TF>(w)}; \
                      return launch_cells(as_stream(stream), op, 0, g->icells, 0, g->jcells, 0, 1, g->icells, g->ijcells); }()
    return MHH_DISPATCH(g, CALL);
#undef CALL
}
