// k_rhs.hip -- operator-level entry points: Diff::exec_viscosity, Diff::exec, the fused advec+diff RHS pass,
// and the max-reductions behind get_cfl / get_dn / check_divergence. gfx950 only.
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include "k_common.h"
#include <wave_reduce.h>   // angle form: the CPU emulation build (tests/emul) overrides it by include path

using namespace mhh;

int mhh_rhs25_march(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, void* stream);   // k_march.hip
int mhh_diff_smag2_march(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, void* stream);   // k_march.hip
int mhh_visc_march(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, const void* th, void* stream);   // k_visc.hip
int mhh_visc_march_rows(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, const void* th, int j0, int j1, void* stream);
int mhh_visc_march_rows2(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, const void* th, int j0, int j1, int j2, int j3, void* stream);
int mhh_rhs25_march_rows2(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, int j2, int j3, void* stream);   // k_march.hip
int mhh_rhs25_march_rows(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, void* stream);   // k_march.hip
int mhh_rhs44_march(const mhh_grid* g, const mhh_fields* f, void* stream);
int mhh_diff4_march(const mhh_grid* g, const mhh_fields* f, void* stream);                                           // k_march4.hip

// =======================================================================================================
// Max reductions (calc_cfl / calc_dnmul / calc_divergence + Master::max). All integrands are |.| >= 0, so the
// IEEE bit pattern orders like an unsigned integer: wave shuffle -> LDS -> one integer atomicMax per block.
// (The reference GPU path seeds its reduction with -FLT_MAX, src/tools.cu:121; the CPU path, our parity
// target, seeds with 0 -- so does this.)
// =======================================================================================================
template<class TF, class Op>
__global__ void __launch_bounds__(BX*BY) max_kernel(const Op op, typename Bits<TF>::U* __restrict__ out,
                                                    int i0, int i1, int j0, int j1, int k0, int k1, int jj, int kk)
{
    const int i = i0 + blockIdx.x*BX + threadIdx.x;
    const int j = j0 + blockIdx.y*BY + threadIdx.y;
    TF m = TF(0);
    if (i < i1 && j < j1)
        for (int k = k0 + blockIdx.z; k < k1; k += gridDim.z)
            m = tmax(m, op(i, j, k, i + j*jj + k*kk));
    block_max_publish<TF, BY>(m, out);
}
template<class TF, class Op>
static int reduce_max(const mhh_grid* g, const Op& op, void* work, double* out, hipStream_t st)
{
    MHH_REQUIRE(work && out, "null work/out");
    using U = typename Bits<TF>::U;
    MHH_HIP_TRY(hipMemsetAsync(work, 0, sizeof(unsigned long long), st));
    const int nz = g->kmax < 32 ? g->kmax : 32;
    dim3 grid((g->imax + BX-1)/BX, (g->jmax + BY-1)/BY, nz);
    hipLaunchKernelGGL((max_kernel<TF, Op>), grid, dim3(BX, BY), 0, st, op, static_cast<U*>(work),
                       g->istart, g->iend, g->jstart, g->jend, g->kstart, g->kend, g->icells, g->ijcells);
    MHH_LAUNCH_CHECK();
    TF h = 0;
    MHH_HIP_TRY(hipMemcpyAsync(&h, work, sizeof(TF), hipMemcpyDeviceToHost, st));
    MHH_HIP_TRY(hipStreamSynchronize(st));
    *out = static_cast<double>(h);
    return MHH_OK;
}
MHH_API unsigned long long mhh_reduce_work_bytes(void) { return 64; }

template<class TF> struct CflOp
{
    GridDev<TF> g; int scheme; const TF* __restrict__ u; const TF* __restrict__ v; const TF* __restrict__ w;
    __device__ TF operator()(int, int, int k, int c) const
    {
        const bool t = (scheme == MHH_ADVEC_4);
        return cfl_cell(scheme, u, v, w, c, g.icells, g.ijcells, k, g.kstart, g.kend, t ? g.dxi_t : g.dxi_d, t ? g.dyi_t : g.dyi_d, g.dzi[k]);
    }
};
MHH_API int mhh_advec_cfl(const mhh_grid* g, int scheme, const void* u, const void* v, const void* w, double dt, void* work, double* cfl_out, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(scheme == MHH_ADVEC_2 || scheme == MHH_ADVEC_2I5 || scheme == MHH_ADVEC_4 || scheme == MHH_ADVEC_2I4 || scheme == MHH_ADVEC_2I62 || scheme == MHH_ADVEC_2I53 || scheme == MHH_ADVEC_4M, "scheme");
    MHH_REQUIRE(u && v && w, "null field");
    double m = 0; int e;
    if (g->dtype == MHH_F64) { CflOp<double> op{make_grid<double>(g), scheme, cp<double>(u), cp<double>(v), cp<double>(w)}; e = reduce_max<double>(g, op, work, &m, as_stream(stream)); if (!e) *cfl_out = m*dt; }
    else { CflOp<float> op{make_grid<float>(g), scheme, cp<float>(u), cp<float>(v), cp<float>(w)}; e = reduce_max<float>(g, op, work, &m, as_stream(stream)); if (!e) *cfl_out = (double)((float)m*(float)dt); }
    return e;
}
template<class TF> struct DnmulOp
{
    GridDev<TF> g; const TF* __restrict__ ev; TF tPrfac_i;
    __device__ TF operator()(int, int, int k, int c) const { return tabs(ev[c]*tPrfac_i*(g.dxidxi_d + g.dyidyi_d + g.dzi[k]*g.dzi[k])); }
};
MHH_API int mhh_smag2_dnmul(const mhh_grid* g, const void* ev, double tPr, void* work, double* out, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(ev != nullptr, "null field");
    if (g->dtype == MHH_F64) { const double t = tPr; DnmulOp<double> op{make_grid<double>(g), cp<double>(ev), 1./(t < 1. ? t : 1.)}; return reduce_max<double>(g, op, work, out, as_stream(stream)); }
    const float t = (float)tPr; DnmulOp<float> op{make_grid<float>(g), cp<float>(ev), 1.f/(t < 1.f ? t : 1.f)};
    return reduce_max<float>(g, op, work, out, as_stream(stream));
}
template<class TF> struct DivOp
{
    GridDev<TF> g; int order; const TF* __restrict__ u; const TF* __restrict__ v; const TF* __restrict__ w;
    const TF* __restrict__ r; const TF* __restrict__ rh;
    __device__ TF operator()(int, int, int k, int c) const
    {
        if (order == 2) return tabs(div2_cell(u, v, w, c, g.icells, g.ijcells, g.dxi_t, g.dyi_t, r[k], rh[k], rh[k+1], g.dzi[k]));
        return tabs(div4_cell(u, v, w, c, g.icells, g.ijcells, g.dxi_d, g.dyi_d, g.dzi4[k]));
    }
};
MHH_API int mhh_pres_check_divergence(const mhh_grid* g, int order, const mhh_fields* f, void* work, double* out, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(order == 2 || order == 4, "order");
    MHH_REQUIRE(f && f->u && f->v && f->w, "null field");
    MHH_REQUIRE(order == 4 || (f->rhoref && f->rhorefh), "rhoref");
    if (g->dtype == MHH_F64) { DivOp<double> op{make_grid<double>(g), order, cp<double>(f->u), cp<double>(f->v), cp<double>(f->w), cp<double>(f->rhoref), cp<double>(f->rhorefh)}; return reduce_max<double>(g, op, work, out, as_stream(stream)); }
    DivOp<float> op{make_grid<float>(g), order, cp<float>(f->u), cp<float>(f->v), cp<float>(f->w), cp<float>(f->rhoref), cp<float>(f->rhorefh)};
    return reduce_max<float>(g, op, work, out, as_stream(stream));
}

// =======================================================================================================
// Diff::exec_viscosity for diff_smag2 (src/diff_smag2.cxx:1046-1188): strain^2 -> (N2) -> evisc -> cyclic fill,
// as ONE stencil pass (the reference stores strain^2 into evisc, reads N2 from a tmp field and rewrites evisc;
// values are identical because every intermediate is rounded to TF exactly where the reference stores it).
// =======================================================================================================

template<class TF>
struct ViscosityOp
{
    GridDev<TF> g; int sm; int neutral; TF* __restrict__ ev;
    const TF* __restrict__ u; const TF* __restrict__ v; const TF* __restrict__ w;
    const TF* __restrict__ dudz; const TF* __restrict__ dvdz; const TF* __restrict__ dbdz; const TF* __restrict__ z0m;
    const TF* __restrict__ N2; const TF* __restrict__ th; const TF* __restrict__ thref; TF grav;
    const TF* __restrict__ mlen0; TF tPr; const TF* __restrict__ mlen2;
    __device__ void operator()(int i, int j, int k, int c) const
    {
        const int ij = i + j*g.icells;
        const bool mo = sm && (k == g.kstart);
        const TF s2 = smag_strain2(u, v, w, c, g.icells, g.ijcells, mo, mo ? dudz[ij] : TF(0), mo ? dvdz[ij] : TF(0),
                                   g.dxi_d, g.dyi_d, g.dzi[k], g.dzhi[k], g.dzhi[k+1]);
        TF n2 = TF(0);
        if (!neutral)
        {
            if (mo) n2 = dbdz[ij];
            else if (N2) n2 = N2[c];
            else n2 = grav/thref[k]*TF(0.5)*(th[c+g.ijcells] - th[c-g.ijcells])*g.dzi[k];
        }
        const TF fac = mlen2 ? mlen2[k] : evisc_mlen2(sm, neutral, mlen0[k], sm ? g.z[k] : TF(0), sm ? z0m[ij] : TF(0));
        ev[c] = evisc_from_mlen2(fac, s2, n2, neutral, tPr);
    }
};
template<class TF>
struct MirrorWallOp2
{
    GridDev<TF> g; TF* __restrict__ ev;
    __device__ void operator()(int i, int j, int, int) const
    {
        const int b = i + j*g.icells + g.kstart*g.ijcells, t = i + j*g.icells + (g.kend-1)*g.ijcells;
        ev[b-g.ijcells] = ev[b];
        ev[t+g.ijcells] = ev[t];
    }
};

// rows [j0, j1) (j0 < 0: the interior, plus the two adjacent ghost rows when p->evisc_ghost_rows), then the wall mirror
// and the east-west wrap over all rows
static int viscosity_rows(const mhh_grid* g, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, void* stream, int j2 = -1, int j3 = -1)
{
    MHH_REQUIRE(f && p && f->evisc && f->u && f->v && f->w && p->mlen0, "null field");
    MHH_REQUIRE(!p->surface_model || (f->dudz && f->dvdz && f->z0m), "surface model inputs");
    const void* th = nullptr;
    if (!p->neutral)
    {
        MHH_REQUIRE(!p->surface_model || f->dbdz, "dbdz");
        if (!p->N2)
        {
            MHH_REQUIRE(p->th_for_N2 >= 0 && p->th_for_N2 < f->nscalars && p->thref, "N2 or (th_for_N2, thref) required");
            th = f->s[p->th_for_N2];
        }
    }
    MHH_REQUIRE(!(p->neutral && !p->surface_model), "neutral + resolved walls: use mhh_smag2_strain2 + mhh_smag2_evisc_neutral");
    MHH_REQUIRE(!p->evisc_ghost_rows || g->jgc >= 2, "evisc_ghost_rows needs jgc >= 2");
    hipStream_t st = as_stream(stream);
    const bool whole = (j0 < 0);
    const int ja = whole ? g->jstart : j0, jb = whole ? g->jend : j1;
    // the k-marching LDS kernel (k_visc.hip), or one thread per cell where it declines
    const int marched = whole ? mhh_visc_march(g, f, p, th, stream) : mhh_visc_march_rows2(g, f, p, th, ja, jb, j2, j3, stream);
#define CALL(TF) [&]{ ViscosityOp<TF> op{make_grid<TF>(g), p->surface_model, p->neutral, mp<TF>(f->evisc), cp<TF>(f->u), cp<TF>(f->v), cp<TF>(f->w), \
                          cp<TF>(f->dudz), cp<TF>(f->dvdz), cp<TF>(f->dbdz), cp<TF>(f->z0m), cp<TF>(p->N2), cp<TF>(th), cp<TF>(p->thref), TF(p->grav), cp<TF>(p->mlen0), TF(p->tPr), cp<TF>(p->mlen2)}; \
                      if (marched < 0) return -marched; \
                      if (!marched) { if (whole) { if (int e = launch_interior(st, op.g, g->kstart, g->kend, op)) return e; } \
                                      else { if (int e = launch_cells(st, op, g->istart, g->iend, ja, jb, g->kstart, g->kend, g->icells, g->ijcells)) return e; \
                                             if (j2 >= 0) if (int e = launch_cells(st, op, g->istart, g->iend, j2, j3, g->kstart, g->kend, g->icells, g->ijcells)) return e; } } \
                      if (whole && p->evisc_ghost_rows) { \
                          if (int e = launch_cells(st, op, g->istart, g->iend, g->jstart-1, g->jstart, g->kstart, g->kend, g->icells, g->ijcells)) return e; \
                          if (int e = launch_cells(st, op, g->istart, g->iend, g->jend, g->jend+1, g->kstart, g->kend, g->icells, g->ijcells)) return e; } \
                      if (!p->surface_model) { MirrorWallOp2<TF> m{op.g, mp<TF>(f->evisc)}; if (int e = launch_cells(st, m, 0, g->icells, 0, g->jcells, 0, 1, g->icells, g->ijcells)) return e; } \
                      return MHH_OK; }()
    if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
    // slab-decomposed: only the local east-west wrap; the caller exchanges the north-south halo
    return mhh_boundary_cyclic(g, f->evisc, g->npy > 1 ? MHH_EDGE_EW : MHH_EDGE_BOTH, stream);
}
MHH_API int mhh_diff_exec_viscosity(const mhh_grid* g, int scheme, const mhh_fields* f, const mhh_diff_params* p, void* stream)
{
    if (int e = check_grid(g)) return e;
    if (scheme != MHH_DIFF_SMAG2) return MHH_OK;          // diff_2 / diff_4: no-op (src/diff_2.h, src/diff_4.h)
    return viscosity_rows(g, f, p, -1, -1, stream);
}
MHH_API int mhh_diff_exec_viscosity_rows(const mhh_grid* g, int scheme, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, void* stream)
{
    if (int e = check_grid(g)) return e;
    if (scheme != MHH_DIFF_SMAG2) return MHH_OK;
    MHH_REQUIRE(j0 >= g->jstart-1 && j0 < j1 && j1 <= g->jend+1, "rows must lie in [jstart-1, jend+1)");
    MHH_REQUIRE((j0 >= g->jstart && j1 <= g->jend) || g->jgc >= 2, "ghost rows need jgc >= 2");
    return viscosity_rows(g, f, p, j0, j1, stream);
}
// two row ranges in ONE launch (+ one wall mirror and one east-west wrap): the two edge strips of a slab once its halos are in
MHH_API int mhh_diff_exec_viscosity_rows2(const mhh_grid* g, int scheme, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, int j2, int j3, void* stream)
{
    if (int e = check_grid(g)) return e;
    if (scheme != MHH_DIFF_SMAG2) return MHH_OK;
    MHH_REQUIRE(j0 >= g->jstart-1 && j0 < j1 && j1 <= j2 && j2 < j3 && j3 <= g->jend+1, "two disjoint, ordered row ranges in [jstart-1, jend+1)");
    MHH_REQUIRE((j0 >= g->jstart && j3 <= g->jend) || g->jgc >= 2, "ghost rows need jgc >= 2");
    return viscosity_rows(g, f, p, j0, j1, stream, j2, j3);
}

// Diff::exec (src/diff_2.cxx:150-180, src/diff_4.cxx:250-300, src/diff_smag2.cxx:939-1043), unfused
MHH_API int mhh_diff_exec(const mhh_grid* g, int scheme, const mhh_fields* f, const mhh_diff_params* p, void* stream)
{
    MHH_REQUIRE(f != nullptr, "fields");
    MHH_REQUIRE(f->nscalars >= 0 && f->nscalars <= MHH_MAX_SCALARS, "nscalars");
    if (scheme == MHH_DIFF_2 || scheme == MHH_DIFF_4)
    {
        const int o = (scheme == MHH_DIFF_2) ? 2 : 4;
        bool uvw_done = false;
        if (o == 4 && g && g->igc >= 3 && g->jgc >= 3 && g->kgc >= 3)     // u, v, w in one pass of the 4th-order marching kernel (diffusive terms only)
        {
            if (int e = check_grid(g)) return e;
            MHH_REQUIRE(f->u && f->v && f->w && f->ut && f->vt && f->wt, "null field");
            const int rc = mhh_diff4_march(g, f, stream);
            if (rc < 0) return -rc;
            uvw_done = (rc == 1);
        }
        if (!uvw_done)
        {
            if (int e = mhh_diff_c(g, o, f->ut, f->u, f->visc, stream)) return e;
            if (int e = mhh_diff_c(g, o, f->vt, f->v, f->visc, stream)) return e;
            if (int e = mhh_diff_w(g, o, f->wt, f->w, f->visc, stream)) return e;
        }
        for (int n=0; n<f->nscalars; ++n)
            if (int e = mhh_diff_c(g, o, f->st[n], f->s[n], f->svisc[n], stream)) return e;
        return MHH_OK;
    }
    MHH_REQUIRE(scheme == MHH_DIFF_SMAG2 && p, "scheme must be 2, 4 or 22 (with params)");
    const int sm = p->surface_model;
    // u, v, w and the first scalar in one pass of the marching kernel with the diffusive terms only (k_march.hip; same bits
    // as the per-field kernels, which MHH_DIFF22_IMPL=cell selects); further scalars per field. Needs the advec_2i5 halo
    // (the tiles are cut for it): other layouts take the per-field kernels.
    const char* impl = getenv("MHH_DIFF22_IMPL");
    if (!(impl && !strcmp(impl, "cell")) && g && g->igc >= 3 && g->jgc >= 3 && g->kgc >= 1 && g->ktot >= 6)
    {
        if (int e = check_grid(g)) return e;
        MHH_REQUIRE(f->u && f->v && f->w && f->ut && f->vt && f->wt && f->evisc && f->rhoref && f->rhorefh, "null field");
        if (sm) MHH_REQUIRE(f->u_fluxbot && f->u_fluxtop && f->v_fluxbot && f->v_fluxtop, "surface fluxes");
        for (int n=0; n<f->nscalars; ++n) MHH_REQUIRE(f->s[n] && f->st[n] && (!sm || (f->s_fluxbot[n] && f->s_fluxtop[n])), "null scalar / scalar surface fluxes");
        mhh_fields fm = *f; fm.nscalars = f->nscalars > 0 ? 1 : 0;
        if (int e = mhh_diff_smag2_march(g, &fm, p, stream)) return e;
        for (int n=1; n<f->nscalars; ++n)
            if (int e = mhh_smag2_diff_c(g, sm, f->st[n], f->s[n], f->evisc, f->s_fluxbot[n], f->s_fluxtop[n], f->rhoref, f->rhorefh, p->tPr, f->svisc[n], stream)) return e;
        return MHH_OK;
    }
    if (int e = mhh_smag2_diff_u(g, sm, f->ut, f->u, f->v, f->w, f->evisc, f->u_fluxbot, f->u_fluxtop, f->rhoref, f->rhorefh, f->visc, stream)) return e;
    if (int e = mhh_smag2_diff_v(g, sm, f->vt, f->u, f->v, f->w, f->evisc, f->v_fluxbot, f->v_fluxtop, f->rhoref, f->rhorefh, f->visc, stream)) return e;
    if (int e = mhh_smag2_diff_w(g, f->wt, f->u, f->v, f->w, f->evisc, f->rhoref, f->rhorefh, f->visc, stream)) return e;
    for (int n=0; n<f->nscalars; ++n)
        if (int e = mhh_smag2_diff_c(g, sm, f->st[n], f->s[n], f->evisc, f->s_fluxbot[n], f->s_fluxtop[n], f->rhoref, f->rhorefh, p->tPr, f->svisc[n], stream)) return e;
    return MHH_OK;
}

// =======================================================================================================
// Fused RHS: advec.exec + diff.exec in one pass. One thread per cell computes every tendency of that cell:
// each input array is read once (through L1/L2) and each tendency is read-modify-written once, in the same
// order of accumulation as the unfused calls.
// =======================================================================================================
template<class TF> struct FieldsDev
{
    const TF* __restrict__ u; const TF* __restrict__ v; const TF* __restrict__ w;
    TF* __restrict__ ut; TF* __restrict__ vt; TF* __restrict__ wt;
    int ns;
    const TF* s[MHH_MAX_SCALARS]; TF* st[MHH_MAX_SCALARS]; TF svisc[MHH_MAX_SCALARS];
    const TF* sfb[MHH_MAX_SCALARS]; const TF* sft[MHH_MAX_SCALARS];
    const TF* __restrict__ ev; const TF* __restrict__ rhoref; const TF* __restrict__ rhorefh;
    const TF* __restrict__ ufb; const TF* __restrict__ uft; const TF* __restrict__ vfb; const TF* __restrict__ vft;
    TF visc, tPr; int sm;
    const TF* __restrict__ bth; const TF* __restrict__ threfh; TF grav; int border;   // folded buoyancy (bth == nullptr: off)
};
template<class TF>
static FieldsDev<TF> make_fields(const mhh_fields* f, const mhh_diff_params* p)
{
    FieldsDev<TF> d;
    d.u = cp<TF>(f->u); d.v = cp<TF>(f->v); d.w = cp<TF>(f->w);
    d.ut = mp<TF>(f->ut); d.vt = mp<TF>(f->vt); d.wt = mp<TF>(f->wt);
    d.ns = f->nscalars;
    for (int n=0; n<MHH_MAX_SCALARS; ++n)
    {
        const bool on = n < f->nscalars;
        d.s[n] = on ? cp<TF>(f->s[n]) : nullptr; d.st[n] = on ? mp<TF>(f->st[n]) : nullptr; d.svisc[n] = on ? TF(f->svisc[n]) : TF(0);
        d.sfb[n] = on ? cp<TF>(f->s_fluxbot[n]) : nullptr; d.sft[n] = on ? cp<TF>(f->s_fluxtop[n]) : nullptr;
    }
    d.ev = cp<TF>(f->evisc); d.rhoref = cp<TF>(f->rhoref); d.rhorefh = cp<TF>(f->rhorefh);
    d.ufb = cp<TF>(f->u_fluxbot); d.uft = cp<TF>(f->u_fluxtop); d.vfb = cp<TF>(f->v_fluxbot); d.vft = cp<TF>(f->v_fluxtop);
    d.visc = TF(f->visc); d.tPr = p ? TF(p->tPr) : TF(1); d.sm = p ? p->surface_model : 0;
    const bool buoy = p && p->buoyancy;
    d.bth = buoy ? cp<TF>(f->s[p->th_for_N2]) : nullptr; d.threfh = buoy ? cp<TF>(p->threfh) : nullptr;
    d.grav = buoy ? TF(p->grav) : TF(0); d.border = buoy ? p->buoyancy : 0;
    return d;
}

// ---- advec_2 + diff_2 --------------------------------------------------------------------------------
template<class TF>
struct Rhs22Op
{
    GridDev<TF> g; FieldsDev<TF> f;
    __device__ void operator()(int, int, int k, int c) const
    {
        const int jj = g.icells, kk = g.ijcells;
        const TF rhk = f.rhorefh[k], rhkp = f.rhorefh[k+1], rk = f.rhoref[k];
        const TF dzi = g.dzi[k], dzhi = g.dzhi[k], dzhip = g.dzhi[k+1];
        {
            TF t = f.ut[c];
            t += advec2_mom(f.u, f.u, f.v, f.w, c, -1, jj, kk, g.dxi_t, g.dyi_t, rhkp, rhk, rk, dzi);
            f.ut[c] = diff2_apply(t, f.u, c, jj, kk, f.visc, g.dxidxi_2, g.dyidyi_2, dzhip, dzhi, dzi);
        }
        {
            TF t = f.vt[c];
            t += advec2_mom(f.v, f.u, f.v, f.w, c, -jj, jj, kk, g.dxi_t, g.dyi_t, rhkp, rhk, rk, dzi);
            f.vt[c] = diff2_apply(t, f.v, c, jj, kk, f.visc, g.dxidxi_2, g.dyidyi_2, dzhip, dzhi, dzi);
        }
        if (k > g.kstart)
        {
            TF t = f.wt[c];
            if (f.bth) t += buoyancy_tend(f.bth, c, kk, f.border, f.grav, f.threfh[k]);
            t += advec2_mom(f.w, f.u, f.v, f.w, c, -kk, jj, kk, g.dxi_t, g.dyi_t, rk, f.rhoref[k-1], rhk, dzhi);
            f.wt[c] = diff2_apply(t, f.w, c, jj, kk, f.visc, g.dxidxi_2, g.dyidyi_2, dzi, g.dzi[k-1], dzhi);
        }
        for (int n=0; n<f.ns; ++n)
        {
            TF t = f.st[n][c];
            t += advec2_s(f.s[n], f.u, f.v, f.w, c, jj, kk, g.dxi_t, g.dyi_t, rhkp, rhk, rk, dzi);
            f.st[n][c] = diff2_apply(t, f.s[n], c, jj, kk, f.svisc[n], g.dxidxi_2, g.dyidyi_2, dzhip, dzhi, dzi);
        }
    }
};

// ---- advec_2i5 + diff_smag2 ----------------------------------------------------------------------------
template<class TF>
struct Rhs25SmagOp
{
    GridDev<TF> g; FieldsDev<TF> f;
    __device__ void operator()(int i, int j, int k, int c) const
    {
        const int jj = g.icells, kk = g.ijcells;
        const int ij = i + j*jj;
        const TF rhk = f.rhorefh[k], rhkp = f.rhorefh[k+1], rk = f.rhoref[k];
        const TF dzi = g.dzi[k], dzhi = g.dzhi[k], dzhip = g.dzhi[k+1];
        const int otc = order_face_c(k+1, g.kstart, g.kend), obc = order_face_c(k, g.kstart, g.kend);
        const bool fb = f.sm && (k == g.kstart), ft = f.sm && (k == g.kend-1);
        {   // u
            const int o = -1;
            TF t = f.ut[c];
            t += advec25_hor(f.u, c, jj, i2(f.u[c+1+o], f.u[c+1]), i2(f.u[c+o], f.u[c]), i2(f.v[c+jj+o], f.v[c+jj]), i2(f.v[c+o], f.v[c]), g.dxi_t, g.dyi_t);
            t += advec25_ver(f.u, c, kk, otc, obc, i2(f.w[c+kk+o], f.w[c+kk]), i2(f.w[c+o], f.w[c]), rhkp, rhk, rk, dzi);
            t += smag_diff_u(f.u, f.v, f.w, f.ev, c, jj, kk, fb, ft, fb ? f.ufb[ij] : TF(0), ft ? f.uft[ij] : TF(0), f.visc, g.dxi_d, g.dyi_d, rhk, rhkp, rk, dzi, dzhi, dzhip);
            f.ut[c] = t;
        }
        {   // v
            const int o = -jj;
            TF t = f.vt[c];
            t += advec25_hor(f.v, c, jj, i2(f.u[c+1+o], f.u[c+1]), i2(f.u[c+o], f.u[c]), i2(f.v[c+jj+o], f.v[c+jj]), i2(f.v[c+o], f.v[c]), g.dxi_t, g.dyi_t);
            t += advec25_ver(f.v, c, kk, otc, obc, i2(f.w[c+kk+o], f.w[c+kk]), i2(f.w[c+o], f.w[c]), rhkp, rhk, rk, dzi);
            t += smag_diff_v(f.u, f.v, f.w, f.ev, c, jj, kk, fb, ft, fb ? f.vfb[ij] : TF(0), ft ? f.vft[ij] : TF(0), f.visc, g.dxi_d, g.dyi_d, rhk, rhkp, rk, dzi, dzhi, dzhip);
            f.vt[c] = t;
        }
        if (k > g.kstart)
        {   // w
            const int o = -kk;
            const TF rkm = f.rhoref[k-1];
            TF t = f.wt[c];
            if (f.bth) t += buoyancy_tend(f.bth, c, kk, f.border, f.grav, f.threfh[k]);
            t += advec25_hor(f.w, c, jj, i2(f.u[c+1+o], f.u[c+1]), i2(f.u[c+o], f.u[c]), i2(f.v[c+jj+o], f.v[c+jj]), i2(f.v[c+o], f.v[c]), g.dxi_t, g.dyi_t);
            t += advec25_ver(f.w, c, kk, order_face_w(k, g.kstart, g.kend), order_face_w(k-1, g.kstart, g.kend),
                             i2(f.w[c+kk+o], f.w[c+kk]), i2(f.w[c+o], f.w[c]), rk, rkm, rhk, dzhi);
            t += smag_diff_w(f.u, f.v, f.w, f.ev, c, jj, kk, f.visc, g.dxi_d, g.dyi_d, rk, rkm, rhk, dzi, g.dzi[k-1], dzhi);
            f.wt[c] = t;
        }
        for (int n=0; n<f.ns; ++n)
        {
            const TF* __restrict__ s = f.s[n];
            TF t = f.st[n][c];
            t += advec25_hor(s, c, jj, f.u[c+1], f.u[c], f.v[c+jj], f.v[c], g.dxi_t, g.dyi_t);
            t += advec25_ver(s, c, kk, otc, obc, f.w[c+kk], f.w[c], rhkp, rhk, rk, dzi);
            t += smag_diff_c(s, f.ev, c, jj, kk, fb, ft, fb ? f.sfb[n][ij] : TF(0), ft ? f.sft[n][ij] : TF(0), f.tPr, f.svisc[n], g.dxidxi_d, g.dyidyi_d, rhk, rhkp, rk, dzi, dzhi, dzhip);
            f.st[n][c] = t;
        }
    }
};

// ---- advec_4 + diff_4 ------------------------------------------------------------------------------------
template<class TF>
struct Rhs44Op
{
    GridDev<TF> g; FieldsDev<TF> f;
    __device__ TF both(TF t, const TF ad[3], const TF df[3]) const
    {
        t -= ad[0]; if (g.dim3) t -= ad[1]; t -= ad[2];
        t += df[0]; if (g.dim3) t += df[1]; t += df[2];
        return t;
    }
    __device__ void operator()(int, int, int k, int c) const
    {
        const int jj = g.icells, kk = g.ijcells;
        const bool bot = (k == g.kstart), top = (k == g.kend-1);
        const TF gc4[4] = {g.dzhi4[k-1], g.dzhi4[k], g.dzhi4[k+1], g.dzhi4[k+2]};
        TF ad[3], df[3];
        advec4_mom(ad, f.u, f.u, f.v, f.w, c, 1, false, jj, kk, bot, top, g.dxi_t, g.dyi_t, g.dzi4[k], g.dim3);
        diff4_cell(df, f.u, c, jj, kk, bot, top, f.visc, g.dxidxi_d, g.dyidyi_d, gc4, g.dzi4[k], g.dim3);
        f.ut[c] = both(f.ut[c], ad, df);
        advec4_mom(ad, f.v, f.u, f.v, f.w, c, jj, false, jj, kk, bot, top, g.dxi_t, g.dyi_t, g.dzi4[k], g.dim3);
        diff4_cell(df, f.v, c, jj, kk, bot, top, f.visc, g.dxidxi_d, g.dyidyi_d, gc4, g.dzi4[k], g.dim3);
        f.vt[c] = both(f.vt[c], ad, df);
        if (k > g.kstart)
        {
            const bool botw = (k == g.kstart+1);
            const TF gw4[4] = {g.dzi4[k-2], g.dzi4[k-1], g.dzi4[k], g.dzi4[k+1]};
            advec4_mom(ad, f.w, f.u, f.v, f.w, c, kk, true, jj, kk, botw, top, g.dxi_t, g.dyi_t, g.dzhi4[k], g.dim3);
            diff4_cell(df, f.w, c, jj, kk, botw, top, f.visc, g.dxidxi_t, g.dyidyi_t, gw4, g.dzhi4[k], g.dim3);
            f.wt[c] = both(f.bth ? f.wt[c] + buoyancy_tend(f.bth, c, kk, f.border, f.grav, f.threfh[k]) : f.wt[c], ad, df);
        }
        for (int n=0; n<f.ns; ++n)
        {
            advec4_s(ad, f.s[n], f.u, f.v, f.w, c, jj, kk, bot, top, g.dxi_t, g.dyi_t, g.dzi4[k], g.dim3);
            diff4_cell(df, f.s[n], c, jj, kk, bot, top, f.svisc[n], g.dxidxi_d, g.dyidyi_d, gc4, g.dzi4[k], g.dim3);
            f.st[n][c] = both(f.st[n][c], ad, df);
        }
    }
};

MHH_API int mhh_rhs_exec(const mhh_grid* g, int advec_scheme, int diff_scheme, const mhh_fields* f, const mhh_diff_params* p, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(f && f->u && f->v && f->w && f->ut && f->vt && f->wt, "null field");
    MHH_REQUIRE(f->nscalars >= 0 && f->nscalars <= MHH_MAX_SCALARS, "nscalars");
    for (int n=0; n<f->nscalars; ++n) MHH_REQUIRE(f->s[n] && f->st[n], "null scalar");
    hipStream_t st = as_stream(stream);
    if (p && p->buoyancy)
    {
        MHH_REQUIRE(p->buoyancy == 2 || p->buoyancy == 4, "buoyancy order must be 2 or 4");
        MHH_REQUIRE(p->th_for_N2 >= 0 && p->th_for_N2 < f->nscalars && p->threfh, "buoyancy needs th_for_N2 and threfh");
        MHH_REQUIRE(g->kgc >= (p->buoyancy == 4 ? 2 : 1), "buoyancy: vertical ghost cells");
    }
    if (advec_scheme != MHH_ADVEC_2I5)
        for (int n=0; n<f->nscalars; ++n) MHH_REQUIRE(!f->s_fluxlimit[n], "fluxlimit_list is an advec_2i5 option (src/advec_2i5.cxx:39)");
    if (advec_scheme == MHH_ADVEC_2 && diff_scheme == MHH_DIFF_2)
    {
        MHH_REQUIRE(f->rhoref && f->rhorefh && g->igc >= 1 && g->jgc >= 1 && g->kgc >= 1, "advec_2+diff_2 inputs");
#define CALL(TF) [&]{ Rhs22Op<TF> op{make_grid<TF>(g), make_fields<TF>(f, p)}; return launch_interior(st, op.g, g->kstart, g->kend, op); }()
        return MHH_DISPATCH(g, CALL);
#undef CALL
    }
    if (advec_scheme == MHH_ADVEC_2I5 && diff_scheme == MHH_DIFF_SMAG2)
    {
        MHH_REQUIRE(p && f->evisc && f->rhoref && f->rhorefh, "advec_2i5+diff_smag2 inputs");
        MHH_REQUIRE(g->igc >= 3 && g->jgc >= 3 && g->kgc >= 1 && g->ktot >= 6, "advec_2i5 needs gc(3,3,1), ktot>=6");
        if (p->surface_model)
        {
            MHH_REQUIRE(f->u_fluxbot && f->u_fluxtop && f->v_fluxbot && f->v_fluxtop, "surface fluxes");
            for (int n=0; n<f->nscalars; ++n) MHH_REQUIRE(f->s_fluxbot[n] && f->s_fluxtop[n], "scalar surface fluxes");
        }
        // default: the k-marching LDS kernel (k_march.hip) for u, v, w and scalar 0; further scalars take the
        // per-field kernels. MHH_RHS25_IMPL=cell selects the one-thread-per-cell fused kernel (A/B measurements).
        const bool use_cell = [] { const char* e = getenv("MHH_RHS25_IMPL"); return e && !strcmp(e, "cell"); }();   // A/B switch, read per call
        bool any_lim = false;
        for (int n=0; n<f->nscalars; ++n) any_lim = any_lim || f->s_fluxlimit[n];
        if (!use_cell || any_lim)
        {
            // the march kernel folds the buoyancy of scalar 0 in (2nd order); otherwise it is added first, on its own
            const mhh_diff_params* pm = p; mhh_diff_params pnb;
            if (p->buoyancy && !(p->buoyancy == 2 && p->th_for_N2 == 0 && !f->s_fluxlimit[0]))
            {
                if (int e = mhh_thermo_dry_buoyancy_tend(g, p->buoyancy, f->wt, f->s[p->th_for_N2], p->threfh, p->grav, stream)) return e;
                pnb = *p; pnb.buoyancy = 0; pm = &pnb;
            }
            // a flux-limited scalar 0 (advec.fluxlimit_list, src/advec_2i5.cxx:921) leaves the fused kernel to u, v, w
            int first = 1;
            if (f->nscalars > 0 && f->s_fluxlimit[0])
            {
                mhh_fields fm = *f; fm.nscalars = 0; first = 0;
                if (int e = mhh_rhs25_march(g, &fm, pm, stream)) return e;
            }
            else if (int e = mhh_rhs25_march(g, f, pm, stream)) return e;
            for (int n=first; n<f->nscalars; ++n)
            {
                if (f->s_fluxlimit[n]) { if (int e = mhh_advec_s_lim(g, f->st[n], f->s[n], f->u, f->v, f->w, f->rhoref, f->rhorefh, stream)) return e; }
                else if (int e = mhh_advec_s(g, MHH_ADVEC_2I5, f->st[n], f->s[n], f->u, f->v, f->w, f->rhoref, f->rhorefh, stream)) return e;
                if (int e = mhh_smag2_diff_c(g, p->surface_model, f->st[n], f->s[n], f->evisc, f->s_fluxbot[n], f->s_fluxtop[n], f->rhoref, f->rhorefh, p->tPr, f->svisc[n], stream)) return e;
            }
            return MHH_OK;
        }
#define CALL(TF) [&]{ Rhs25SmagOp<TF> op{make_grid<TF>(g), make_fields<TF>(f, p)}; return launch_interior(st, op.g, g->kstart, g->kend, op); }()
        return MHH_DISPATCH(g, CALL);
#undef CALL
    }
    if (advec_scheme == MHH_ADVEC_4 && diff_scheme == MHH_DIFF_4)
    {
        MHH_REQUIRE(g->igc >= 3 && g->jgc >= 3 && g->kgc >= 3, "4th order needs gc(3,3,3)");
        // default: the k-marching LDS kernel (k_march4.hip) for u, v, w where the rows allow LDS-DMA; scalars and the folded
        // buoyancy then take their own kernels (same order of accumulation). MHH_RHS44_IMPL=cell selects the cell kernel.
        {
            const mhh_fields* fq = f;
            const bool buoy = p && p->buoyancy;
            // probe the layout first so that the buoyancy term is added exactly once, and before advection
            const char* e = getenv("MHH_RHS44_IMPL");
            const bool can = !(e && !strcmp(e, "cell"));
            if (can)
            {
                if (buoy) if (int e2 = mhh_thermo_dry_buoyancy_tend(g, p->buoyancy, f->wt, f->s[p->th_for_N2], p->threfh, p->grav, stream)) return e2;
                const int m = mhh_rhs44_march(g, fq, stream);
                if (m < 0) return -m;
                MHH_REQUIRE(m == 1, "internal: rhs44 march declined a layout it was probed for");
                for (int n=0; n<f->nscalars; ++n)
                {
                    if (int e2 = mhh_advec_s(g, MHH_ADVEC_4, f->st[n], f->s[n], f->u, f->v, f->w, f->rhoref, f->rhorefh, stream)) return e2;
                    if (int e2 = mhh_diff_c(g, 4, f->st[n], f->s[n], f->svisc[n], stream)) return e2;
                }
                return MHH_OK;
            }
        }
#define CALL(TF) [&]{ Rhs44Op<TF> op{make_grid<TF>(g), make_fields<TF>(f, p)}; return launch_interior(st, op.g, g->kstart, g->kend, op); }()
        return MHH_DISPATCH(g, CALL);
#undef CALL
    }
    // any other pair of valid schemes: the two operator calls, in the reference's order (same bits as calling them directly)
    if (p && p->buoyancy)
        if (int e = mhh_thermo_dry_buoyancy_tend(g, p->buoyancy, f->wt, f->s[p->th_for_N2], p->threfh, p->grav, stream)) return e;
    if (int e = mhh_advec_exec(g, advec_scheme, f, stream)) return e;
    return mhh_diff_exec(g, diff_scheme, f, p, stream);
}

// The (advec_2i5, diff_smag2) pass over the rows [j0, j1) only: the slab driver updates the rows that need no
// north-south halo while the halos travel and the edge rows afterwards. u, v, w and at most one scalar (in the kernel).
MHH_API int mhh_rhs_exec_rows(const mhh_grid* g, int advec_scheme, int diff_scheme, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(advec_scheme == MHH_ADVEC_2I5 && diff_scheme == MHH_DIFF_SMAG2, "row-wise pass: (advec_2i5, diff_smag2) only");
    MHH_REQUIRE(f && p && f->u && f->v && f->w && f->ut && f->vt && f->wt && f->evisc && f->rhoref && f->rhorefh, "null field");
    MHH_REQUIRE(f->nscalars >= 0 && f->nscalars <= 1 && (f->nscalars == 0 || (f->s[0] && f->st[0] && !f->s_fluxlimit[0])), "row-wise pass: at most one, unlimited scalar");
    MHH_REQUIRE(!p->buoyancy || (p->buoyancy == 2 && p->th_for_N2 == 0 && f->nscalars == 1 && p->threfh), "row-wise pass: buoyancy only as the in-kernel 2nd-order form of scalar 0");
    MHH_REQUIRE(g->igc >= 3 && g->jgc >= 3 && g->kgc >= 1 && g->ktot >= 6, "advec_2i5 needs gc(3,3,1), ktot>=6");
    MHH_REQUIRE(j0 >= g->jstart && j0 < j1 && j1 <= g->jend, "rows must lie in [jstart, jend)");
    if (p->surface_model)
    {
        MHH_REQUIRE(f->u_fluxbot && f->u_fluxtop && f->v_fluxbot && f->v_fluxtop, "surface fluxes");
        if (f->nscalars) MHH_REQUIRE(f->s_fluxbot[0] && f->s_fluxtop[0], "scalar surface fluxes");
    }
    return mhh_rhs25_march_rows(g, f, p, j0, j1, stream);
}
// the same over two row ranges in one launch (the two edge strips of a slab)
MHH_API int mhh_rhs_exec_rows2(const mhh_grid* g, int advec_scheme, int diff_scheme, const mhh_fields* f, const mhh_diff_params* p, int j0, int j1, int j2, int j3, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(advec_scheme == MHH_ADVEC_2I5 && diff_scheme == MHH_DIFF_SMAG2, "row-wise pass: (advec_2i5, diff_smag2) only");
    MHH_REQUIRE(f && p && f->u && f->v && f->w && f->ut && f->vt && f->wt && f->evisc && f->rhoref && f->rhorefh, "null field");
    MHH_REQUIRE(f->nscalars >= 0 && f->nscalars <= 1 && (f->nscalars == 0 || (f->s[0] && f->st[0] && !f->s_fluxlimit[0])), "row-wise pass: at most one, unlimited scalar");
    MHH_REQUIRE(!p->buoyancy || (p->buoyancy == 2 && p->th_for_N2 == 0 && f->nscalars == 1 && p->threfh), "row-wise pass: buoyancy only as the in-kernel 2nd-order form of scalar 0");
    MHH_REQUIRE(g->igc >= 3 && g->jgc >= 3 && g->kgc >= 1 && g->ktot >= 6, "advec_2i5 needs gc(3,3,1), ktot>=6");
    MHH_REQUIRE(j0 >= g->jstart && j0 < j1 && j1 <= j2 && j2 < j3 && j3 <= g->jend, "two disjoint, ordered row ranges in [jstart, jend)");
    if (p->surface_model)
    {
        MHH_REQUIRE(f->u_fluxbot && f->u_fluxtop && f->v_fluxbot && f->v_fluxtop, "surface fluxes");
        if (f->nscalars) MHH_REQUIRE(f->s_fluxbot[0] && f->s_fluxtop[0], "scalar surface fluxes");
    }
    return mhh_rhs25_march_rows2(g, f, p, j0, j1, j2, j3, stream);
}
