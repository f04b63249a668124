// k_pres.hip -- Pres_2 / Pres_4: FFT-based Poisson solver on one GPU (gfx950), rocFFT for the horizontal
// transforms.
//
// Data flow (Pres::exec, src/pres_2.cxx:66-94, src/pres_4.cxx:64-140):
//   input   : cyclic fill of ut (E-W), vt (N-S); divergence -> packed p (imax*jmax*kmax reals, no ghosts)
//   forward : rocFFT batched 2-D real-to-complex over every k plane -> spec[ktot][jtot][itot/2+1]
//   solve   : one thread per (kx,ky) column: Thomas (pres_2) / banded LU (pres_4) in k, on re and im at once
//   backward: rocFFT 2-D complex-to-real (unnormalised)
//   unpack  : /jtot /itot, write ghosted p incl. vertical ghost rows and the periodic halo in one kernel
//   output  : ut,vt,wt -= grad p
// The reference works on FFTW half-complex *real* coefficients; the complex formulation solves the same
// real tridiagonal system per wave-number pair (the matrix is real and depends only on
// bmati[kx]+bmatj[ky]), so results agree to rounding of the transform (DESIGN.md "Parity").
#include <vector>
#include <map>
#include <mutex>
#include <rocfft/rocfft.h>
#include "fft_lifetime.h"
#include <cstring>
#include <cstdlib>
#include "k_common.h"

using namespace mhh;

#define MHH_FFT_TRY(expr) do { rocfft_status s_ = (expr); if (s_ != rocfft_status_success) { \
    mhh::set_error("FFT error: %s returned %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__); return MHH_EFFT; } } while (0)

struct mhh_pres_plan
{
    int order = 0, dtype = 0;
    int itot = 0, jtot = 0, ktot = 0, nxh = 0;
    int nxp = 0;                 // row pitch of the spectral array in complex elements (>= nxh)
    size_t esz = 8;
    // coefficient tables (device), element type = dtype
    void* bmati = nullptr; void* bmatj = nullptr;
    void* a = nullptr; void* c = nullptr; void* dz = nullptr; void* rhoref = nullptr;  // pres_2
    void* m[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};     // pres_4
    // buffers
    void* packed = nullptr;      // imax*jmax*kmax reals
    void* spec = nullptr;        // nxp*jtot*ktot complex (rows of nxh modes at pitch nxp)
    void* work = nullptr;        // scratch of the k-sweep: work3d (pres_2) / 7 band arrays + rhs (pres_4)
    rocfft_plan fwd = nullptr, bwd = nullptr;
    rocfft_execution_info fwd_info = nullptr, bwd_info = nullptr;
    // fused form (mhh_pres_exec): Pres::input rides in the forward transform as its load callback, the unpack in the inverse
    // transform as its store callback; cb_data = device copy of the PresCb record the callbacks read
    rocfft_execution_info fwd_info_cb = nullptr, bwd_info_cb = nullptr;
    void* cb_data = nullptr; bool cb_ready = false;
    void* fwd_wb = nullptr; void* bwd_wb = nullptr;
    bool fft_setup = false;
    // the three-kernel form with the transforms in LDS (pres_lds.h): twiddle tables, w3 in its [k][kx][ky] layout
    void* tx = nullptr; void* ty = nullptr; void* w3l = nullptr;
    void* a3l = nullptr; void* itw = nullptr; int ksplit = 0; bool tw_ok = false;     // the two-blocks-per-column form of the y stage (pres_2)
    bool lds_ok = false;
    // mhh_pres_exec_rk: the Runge-Kutta sub-step of u, v, w rides in the kernel that stores the corrected tendencies
    bool rk_on = false; double rk_cA = 0, rk_cB = 0, rk_dt = 0; void* rk_u = nullptr; void* rk_v = nullptr; void* rk_w = nullptr;
};


// naturally aligned (16 bytes in fp64): one ds_read_b128 / global_load_dwordx4 per number instead of two 8-byte halves
template<class TF> struct alignas(2*sizeof(TF)) C2 { TF x, y; };
#include "pres_lds.h"
#include "pres_lds4.h"

// ---- host-side coefficient tables (Pres_2::set_values src/pres_2.cxx:125-153; Pres_4::set_values src/pres_4.cxx:179-252)
template<class TF>
static void host_bmat(int order, const mhh_grid* g, std::vector<TF>& bi, std::vector<TF>& bj)
{
    const int itot = g->itot, jtot = g->jtot;
    const TF dx = TF(g->dx), dy = TF(g->dy);
    const TF dxidxi = 1./(dx*dx), dyidyi = 1./(dy*dy);
    const TF pi = std::acos(-1.);
    bi.resize(itot); bj.resize(jtot);
    if (order == 2)
    {
        for (int j=0; j<jtot/2+1; ++j) bj[j] = 2. * (std::cos(2.*pi*(TF)j/(TF)jtot)-1.) * dyidyi;
        for (int i=0; i<itot/2+1; ++i) bi[i] = 2. * (std::cos(2.*pi*(TF)i/(TF)itot)-1.) * dxidxi;
    }
    else
    {
        for (int j=0; j<jtot/2+1; j++)
            bj[j] = ( 2.* (1./576.) * std::cos(6.*pi*(double)j/(double)jtot) - 2.* (54./576.) * std::cos(4.*pi*(double)j/(double)jtot)
                    + 2.* (783./576.) * std::cos(2.*pi*(double)j/(double)jtot) - (1460./576.) ) * dyidyi;
        for (int i=0; i<itot/2+1; i++)
            bi[i] = ( 2.* (1./576.) * std::cos(6.*pi*(double)i/(double)itot) - 2.* (54./576.) * std::cos(4.*pi*(double)i/(double)itot)
                    + 2.* (783./576.) * std::cos(2.*pi*(double)i/(double)itot) - (1460./576.) ) * dxidxi;
    }
    for (int j=jtot/2+1; j<jtot; ++j) bj[j] = bj[jtot-j];
    for (int i=itot/2+1; i<itot; ++i) bi[i] = bi[itot-i];
}

template<class TF>
static int upload(void** dst, const std::vector<TF>& v)
{
    MHH_HIP_TRY(hipMalloc(dst, v.size()*sizeof(TF)));
    MHH_HIP_TRY(hipMemcpy(*dst, v.data(), v.size()*sizeof(TF), hipMemcpyHostToDevice));
    return MHH_OK;
}

template<class TF>
static int plan_tables(mhh_pres_plan* P, const mhh_grid* g, const void* hdz, const void* hdzhi, const void* hdzi4, const void* hdzhi4,
                       const void* hrho, const void* hrhoh)
{
    std::vector<TF> bi, bj;
    host_bmat<TF>(P->order, g, bi, bj);
    if (int e = upload(&P->bmati, bi)) return e;
    if (int e = upload(&P->bmatj, bj)) return e;
    const int kmax = g->kmax, kgc = g->kgc, kstart = g->kstart;
    if (P->order == 2)
    {
        const TF* dz = cp<TF>(hdz); const TF* dzhi = cp<TF>(hdzhi); const TF* rhoh = cp<TF>(hrhoh); const TF* rho = cp<TF>(hrho);
        std::vector<TF> a(kmax), c(kmax), dzk(kmax), rk(kmax);
        for (int k=0; k<kmax; ++k)
        {
            a[k] = dz[k+kgc] * rhoh[k+kgc  ]*dzhi[k+kgc  ];
            c[k] = dz[k+kgc] * rhoh[k+kgc+1]*dzhi[k+kgc+1];
            dzk[k] = dz[k+kgc]; rk[k] = rho[k+kgc];
        }
        if (int e = upload(&P->a, a)) return e;
        if (int e = upload(&P->c, c)) return e;
        if (int e = upload(&P->dz, dzk)) return e;
        if (int e = upload(&P->rhoref, rk)) return e;
    }
    else
    {
        const TF* dzi4 = cp<TF>(hdzi4); const TF* h = cp<TF>(hdzhi4);
        std::vector<TF> m1(kmax), m2(kmax), m3(kmax), m4(kmax), m5(kmax), m6(kmax), m7(kmax);
        int k = 0, kc = kstart;
        m1[k] = 0.;
        m2[k] = (1./576.) * (               -  27.*h[kc]                            ) * dzi4[kc];
        m3[k] = (1./576.) * ( -1.*h[kc+1] + 729.*h[kc] +  27.*h[kc+1]               ) * dzi4[kc];
        m4[k] = (1./576.) * ( 27.*h[kc+1] - 729.*h[kc] - 729.*h[kc+1] -  1.*h[kc+2] ) * dzi4[kc];
        m5[k] = (1./576.) * (-27.*h[kc+1] +  27.*h[kc] + 729.*h[kc+1] + 27.*h[kc+2] ) * dzi4[kc];
        m6[k] = (1./576.) * (  1.*h[kc+1]              -  27.*h[kc+1] - 27.*h[kc+2] ) * dzi4[kc];
        m7[k] = (1./576.) * (                                         +  1.*h[kc+2] ) * dzi4[kc];
        for (k=1; k<kmax-1; k++)
        {
            kc = kstart+k;
            m1[k] = (1./576.) * (   1.*h[kc-1]                                           ) * dzi4[kc];
            m2[k] = (1./576.) * ( -27.*h[kc-1] -  27.*h[kc]                              ) * dzi4[kc];
            m3[k] = (1./576.) * (  27.*h[kc-1] + 729.*h[kc] +  27.*h[kc+1]               ) * dzi4[kc];
            m4[k] = (1./576.) * (  -1.*h[kc-1] - 729.*h[kc] - 729.*h[kc+1] -  1.*h[kc+2] ) * dzi4[kc];
            m5[k] = (1./576.) * (              +  27.*h[kc] + 729.*h[kc+1] + 27.*h[kc+2] ) * dzi4[kc];
            m6[k] = (1./576.) * (                           -  27.*h[kc+1] - 27.*h[kc+2] ) * dzi4[kc];
            m7[k] = (1./576.) * (                                          +  1.*h[kc+2] ) * dzi4[kc];
        }
        k = kmax-1; kc = kstart+k;
        m1[k] = (1./576.) * (   1.*h[kc-1]                                         ) * dzi4[kc];
        m2[k] = (1./576.) * ( -27.*h[kc-1] -  27.*h[kc]                +  1.*h[kc] ) * dzi4[kc];
        m3[k] = (1./576.) * (  27.*h[kc-1] + 729.*h[kc] +  27.*h[kc+1] - 27.*h[kc] ) * dzi4[kc];
        m4[k] = (1./576.) * (  -1.*h[kc-1] - 729.*h[kc] - 729.*h[kc+1] + 27.*h[kc] ) * dzi4[kc];
        m5[k] = (1./576.) * (              +  27.*h[kc] + 729.*h[kc+1] -  1.*h[kc] ) * dzi4[kc];
        m6[k] = (1./576.) * (                           -  27.*h[kc+1]             ) * dzi4[kc];
        m7[k] = 0.;
        std::vector<TF>* mm[7] = {&m1, &m2, &m3, &m4, &m5, &m6, &m7};
        for (int n=0; n<7; ++n) if (int e = upload(&P->m[n], *mm[n])) return e;
    }
    return MHH_OK;
}

static int make_fft(mhh_pres_plan* P, bool forward, rocfft_plan* plan, rocfft_execution_info* info, void** wb)
{
    const size_t lengths[2] = {(size_t)P->itot, (size_t)P->jtot};
    const size_t rstr[2] = {1, (size_t)P->itot};
    const size_t cstr[2] = {1, (size_t)P->nxp};
    const size_t rdist = (size_t)P->itot*P->jtot, cdist = (size_t)P->nxp*P->jtot;
    const size_t off[2] = {0, 0};
    rocfft_plan_description d = nullptr;
    MHH_FFT_TRY(rocfft_plan_description_create(&d));
    const size_t ndim = (P->jtot > 1) ? 2 : 1;
    if (forward)
        MHH_FFT_TRY(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, off, off, ndim, rstr, rdist, ndim, cstr, cdist));
    else
        MHH_FFT_TRY(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, off, off, ndim, cstr, cdist, ndim, rstr, rdist));
    MHH_FFT_TRY(rocfft_plan_create(plan, rocfft_placement_notinplace, forward ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse,
                                   P->dtype == MHH_F64 ? rocfft_precision_double : rocfft_precision_single, ndim, lengths, (size_t)P->ktot, d));
    MHH_FFT_TRY(rocfft_plan_description_destroy(d));
    MHH_FFT_TRY(rocfft_execution_info_create(info));
    size_t wbs = 0;
    MHH_FFT_TRY(rocfft_plan_get_work_buffer_size(*plan, &wbs));
    if (wbs)
    {
        MHH_HIP_TRY(hipMalloc(wb, wbs));
        MHH_FFT_TRY(rocfft_execution_info_set_work_buffer(*info, *wb, wbs));
    }
    return MHH_OK;
}

MHH_API void mhh_pres_plan_destroy(mhh_pres_plan* P)
{
    if (!P) return;
    if (P->fwd) rocfft_plan_destroy(P->fwd);
    if (P->bwd) rocfft_plan_destroy(P->bwd);
    if (P->fwd_info) rocfft_execution_info_destroy(P->fwd_info);
    if (P->bwd_info) rocfft_execution_info_destroy(P->bwd_info);
    if (P->fwd_info_cb) rocfft_execution_info_destroy(P->fwd_info_cb);
    if (P->bwd_info_cb) rocfft_execution_info_destroy(P->bwd_info_cb);
    void* bufs[] = {P->tx, P->ty, P->w3l, P->a3l, P->itw, P->cb_data, P->bmati, P->bmatj, P->a, P->c, P->dz, P->rhoref, P->packed, P->spec, P->work, P->fwd_wb, P->bwd_wb,
                    P->m[0], P->m[1], P->m[2], P->m[3], P->m[4], P->m[5], P->m[6]};
    for (void* b : bufs) if (b) (void)hipFree(b);
    delete P;
}

static int hdma_factor(mhh_pres_plan* P);
static int tdma_factor(mhh_pres_plan* P);
static int pres_cb_setup(mhh_pres_plan* P);
static int pres_lds_setup(mhh_pres_plan* P, const mhh_grid* g);
MHH_API int mhh_pres_plan_create(const mhh_grid* g, int order, const void* host_dz, const void* host_dzhi, const void* host_dzi4, const void* host_dzhi4,
                                 const void* host_rhoref, const void* host_rhorefh, mhh_pres_plan** out)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(out != nullptr, "out");
    MHH_REQUIRE(order == 2 || order == 4, "order must be 2 or 4");
    MHH_REQUIRE(g->npy == 1 && g->imax == g->itot && g->jmax == g->jtot, "single-GPU plan: use the slab driver for npy > 1");
    if (order == 2) MHH_REQUIRE(host_dz && host_dzhi && host_rhoref && host_rhorefh && g->kgc >= 1 && g->igc >= 1 && g->jgc >= 1, "pres_2 inputs");
    else            MHH_REQUIRE(host_dzi4 && host_dzhi4 && g->kgc >= 2 && g->igc >= 2 && g->jgc >= 2 && g->kmax >= 4, "pres_4 inputs");
    mhh_pres_plan* P = new mhh_pres_plan();
    P->order = order; P->dtype = g->dtype; P->itot = g->itot; P->jtot = g->jtot; P->ktot = g->ktot; P->nxh = g->itot/2 + 1;
    {   // row pitch of the spectral array: itot/2+1 complex numbers is never a whole number of 128-byte lines, and the column
        // tiles of rocFFT's y pass then straddle lines (512^3 fp64: 0.72 -> 0.48 ms per pass with rows padded to 264).
        // MHH_PRES_PITCH=n pads to a multiple of n elements instead (1: no padding) for A/B runs.
        const char* pe = getenv("MHH_PRES_PITCH");
        const int al = pe ? atoi(pe) : 128 / (2*((g->dtype == MHH_F64) ? 8 : 4));
        P->nxp = (al > 1) ? (P->nxh + al-1)/al*al : P->nxh;
    }
    P->esz = (g->dtype == MHH_F64) ? 8 : 4;
    int e = (g->dtype == MHH_F64) ? plan_tables<double>(P, g, host_dz, host_dzhi, host_dzi4, host_dzhi4, host_rhoref, host_rhorefh)
                                  : plan_tables<float>(P, g, host_dz, host_dzhi, host_dzi4, host_dzhi4, host_rhoref, host_rhorefh);
    const size_t nreal = (size_t)g->itot*g->jtot*g->ktot, ncol = (size_t)P->nxp*g->jtot;
    const size_t nwork = (order == 2) ? ncol*g->ktot : ncol*(g->ktot+4)*7;   // pres_4: the 7 factored bands
    if (!e) { hipError_t h = hipMalloc(&P->packed, nreal*P->esz); if (h != hipSuccess) { set_error("hipMalloc packed: %s", hipGetErrorString(h)); e = MHH_ENOMEM; } }
    if (!e) { hipError_t h = hipMalloc(&P->spec, ncol*g->ktot*2*P->esz); if (h != hipSuccess) { set_error("hipMalloc spec: %s", hipGetErrorString(h)); e = MHH_ENOMEM; } }
    if (!e) { hipError_t h = hipMalloc(&P->work, nwork*P->esz); if (h != hipSuccess) { set_error("hipMalloc work: %s", hipGetErrorString(h)); e = MHH_ENOMEM; } }
    if (!e)
    {
        fft_acquire();
        P->fft_setup = true;
        e = make_fft(P, true, &P->fwd, &P->fwd_info, &P->fwd_wb);
        if (!e) e = make_fft(P, false, &P->bwd, &P->bwd_info, &P->bwd_wb);
        if (!e) e = pres_cb_setup(P);
    }
    if (!e && order == 4) e = hdma_factor(P);
    if (!e && order == 2) e = tdma_factor(P);
    if (!e) e = pres_lds_setup(P, g);
    if (e) { mhh_pres_plan_destroy(P); return e; }
    *out = P;
    return MHH_OK;
}

// =======================================================================================================
// input (src/pres_2.cxx:156-196, src/pres_4.cxx:256-317)
// =======================================================================================================
template<class TF>
struct PresInOp
{
    GridDev<TF> g; int order; TF* __restrict__ p;
    const TF* __restrict__ u; const TF* __restrict__ v; const TF* __restrict__ w;
    const TF* __restrict__ ut; const TF* __restrict__ vt; const TF* __restrict__ wt;
    const TF* __restrict__ rhoref; const TF* __restrict__ rhorefh; TF dti2, dti4;
    __device__ void operator()(int i, int j, int k, int c) const
    {
        const size_t cp_ = (size_t)(i-g.igc) + (size_t)(j-g.jgc)*g.imax + (size_t)(k-g.kgc)*g.imax*g.jmax;
        if (order == 2) p[cp_] = pres2_in(u, v, w, ut, vt, wt, c, g.icells, g.ijcells, g.dxi_t, g.dyi_t, dti2, rhoref[k], rhorefh[k], rhorefh[k+1], g.dzi[k]);
        else            p[cp_] = pres4_in(u, v, w, ut, vt, wt, c, g.icells, g.ijcells, g.dxi_d, g.dyi_d, dti4, g.dzi4[k], g.dim3);
    }
};
// wt ghost rows of pres_4 (src/pres_4.cxx:290-303): wt[kstart-1] = -wt[kstart+1], wt[kend+1] = -wt[kend-1] over the interior columns
template<class TF>
struct WtGhostOp
{
    GridDev<TF> g; TF* __restrict__ wt;
    __device__ void operator()(int i, int j, int, int) const
    {
        const int b = i + j*g.icells + g.kstart*g.ijcells, t = i + j*g.icells + g.kend*g.ijcells;
        wt[b-g.ijcells] = -wt[b+g.ijcells];
        wt[t+g.ijcells] = -wt[t-g.ijcells];
    }
};

// what Pres::input does before its stencil: east-west halo of ut, north-south halo of vt, mirrored wt ghost rows (pres_4)
static int pres_input_halos(const mhh_grid* g, int order, const mhh_fields* f, void* stream)
{
    if (int e = mhh_boundary_cyclic(g, f->ut, MHH_EDGE_EW, stream)) return e;
    if (g->npy == 1 && (order == 2 || g->jtot != 1))
        if (int e = mhh_boundary_cyclic(g, f->vt, MHH_EDGE_NS, stream)) return e;
    if (order == 4)
    {
#define CALL(TF) [&]{ WtGhostOp<TF> wg{make_grid<TF>(g), mp<TF>(f->wt)}; return launch_cells(as_stream(stream), wg, g->istart, g->iend, g->jstart, g->jend, 0, 1, g->icells, g->ijcells); }()
        if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
    }
    return MHH_OK;
}
// plan-free form, also used by the slab-decomposed driver (there the north-south halo of vt is the caller's exchange)
MHH_API int mhh_pres_input_packed(const mhh_grid* g, int order, const mhh_fields* f, double dt, void* p_packed, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(f && f->u && f->v && f->w && f->ut && f->vt && f->wt && p_packed, "null field");
    MHH_REQUIRE(order == 2 || order == 4, "order");
    MHH_REQUIRE(dt > 0., "dt");
    MHH_REQUIRE(order == 4 || (f->rhoref && f->rhorefh), "rhoref");
    if (int e = pres_input_halos(g, order, f, stream)) return e;
    hipStream_t st = as_stream(stream);
#define CALL(TF) [&]{ GridDev<TF> gd = make_grid<TF>(g); \
        PresInOp<TF> op{gd, order, mp<TF>(p_packed), cp<TF>(f->u), cp<TF>(f->v), cp<TF>(f->w), cp<TF>(f->ut), cp<TF>(f->vt), cp<TF>(f->wt), \
                        cp<TF>(f->rhoref), cp<TF>(f->rhorefh), TF(1.)/TF(dt), TF(1./TF(dt))}; \
        return launch_interior(st, gd, g->kstart, g->kend, op); }()
    return MHH_DISPATCH(g, CALL);
#undef CALL
}
MHH_API int mhh_pres_input(mhh_pres_plan* P, const mhh_grid* g, const mhh_fields* f, double dt, void* p_packed, void* stream)
{
    MHH_REQUIRE(P != nullptr, "plan");
    MHH_REQUIRE(g && P->dtype == g->dtype && P->itot == g->itot && P->jtot == g->jtot && P->ktot == g->ktot, "plan/grid mismatch");
    return mhh_pres_input_packed(g, P->order, f, dt, p_packed ? p_packed : P->packed, stream);
}

// =======================================================================================================
// spectral solve, pres_2: Thomas algorithm per (kx,ky) column (src/pres_2.cxx:289-330 matrix, :202-263 tdma)
// =======================================================================================================
// The pivots w2 and the eliminated upper diagonal w3 depend on the grid and (kx, ky) only (the reference regenerates them
// in every solve, src/pres_2.cxx:213-249): w3 is factored ONCE at plan creation by the same recurrence (tdma_factor_kernel)
// and kept (one real per spectral element); the forward sweep of a solve runs the cheap recurrence in registers and
// touches no scratch at all, the backward sweep reads w3 -- one array write per solve less than the reference's form.
// Both sweeps work in batches of U levels: the U right-hand sides (and w3 values) of a batch are loaded before its
// dependent chain starts, so every lane keeps U loads in flight instead of one.
template<class TF>
__device__ __forceinline__ TF tdma_diag(const TF* __restrict__ a, const TF* __restrict__ c, const TF* __restrict__ dz, const TF* __restrict__ rho,
                                        TF bm, bool mean, int k, int kmax)
{
    const TF dz2 = dz[k]*dz[k];
    TF b = dz2 * rho[k]*bm - (a[k]+c[k]);
    if (k == 0) b += a[0];
    if (k == kmax-1) { if (mean) b -= c[k]; else b += c[k]; }
    return b;
}
template<class TF>
__global__ void __launch_bounds__(64) tdma_factor_kernel(TF* __restrict__ work3d, const TF* __restrict__ bmati, const TF* __restrict__ bmatj,
                                                         const TF* __restrict__ a, const TF* __restrict__ c, const TF* __restrict__ dz, const TF* __restrict__ rho,
                                                         int nxh, int nxp, int jtot, int kmax)
{
    const size_t ncol = (size_t)nxp*jtot, col = (size_t)blockIdx.x*64 + threadIdx.x;
    if (col >= ncol) return;
    const int ky = (int)(col / nxp), kx = (int)(col - (size_t)ky*nxp);
    if (kx >= nxh) return;
    const TF bm = bmati[kx] + bmatj[ky];
    const bool mean = (kx == 0 && ky == 0);
    TF w2 = tdma_diag(a, c, dz, rho, bm, mean, 0, kmax);
    work3d[col] = TF(0);
    for (int k=1; k<kmax; ++k)
    {
        const TF w3 = c[k-1] / w2;
        work3d[col + (size_t)k*ncol] = w3;
        w2 = tdma_diag(a, c, dz, rho, bm, mean, k, kmax) - a[k]*w3;
    }
}
template<class TF>
__global__ void __launch_bounds__(64) tdma_kernel(C2<TF>* __restrict__ p, const TF* __restrict__ work3d,
                                                  const TF* __restrict__ bmati, const TF* __restrict__ bmatj,
                                                  const TF* __restrict__ a, const TF* __restrict__ c, const TF* __restrict__ dz, const TF* __restrict__ rho,
                                                  int nxh, int nxp, int jtot, int kmax)
{
    constexpr int U = 8;
    // one thread per (kx, ky) column, threads running over the flattened spectral plane (row pitch nxp)
    const size_t ncol = (size_t)nxp*jtot, col = (size_t)blockIdx.x*64 + threadIdx.x;
    if (col >= ncol) return;
    const int ky = (int)(col / nxp), kx = (int)(col - (size_t)ky*nxp);
    if (kx >= nxh) return;
    const TF bm = bmati[kx] + bmatj[ky];
    const bool mean = (kx == 0 && ky == 0);
    TF w2 = TF(1); C2<TF> pp{TF(0), TF(0)};
    for (int k0=0; k0<kmax; k0+=U)
    {
        C2<TF> q[U];
#pragma unroll
        for (int n=0; n<U; ++n) if (k0+n < kmax) q[n] = p[col + (size_t)(k0+n)*ncol];
#pragma unroll
        for (int n=0; n<U; ++n)
        {
            const int k = k0+n;
            if (k < kmax)
            {
                const TF dz2 = dz[k]*dz[k];
                const TF b = tdma_diag(a, c, dz, rho, bm, mean, k, kmax);
                C2<TF> r = q[n];
                r.x = dz2 * r.x; r.y = dz2 * r.y;
                if (k == 0) w2 = b;
                else
                {
                    const TF w3 = c[k-1] / w2;
                    w2 = b - a[k]*w3;
                    r.x -= a[k]*pp.x; r.y -= a[k]*pp.y;
                }
                r.x /= w2; r.y /= w2;
                q[n] = r; pp = r;
            }
        }
#pragma unroll
        for (int n=0; n<U; ++n) if (k0+n < kmax) p[col + (size_t)(k0+n)*ncol] = q[n];
    }
    // back substitution, top batch first: levels kmax-2 .. 0
    for (int k1=kmax-2; k1>=0; k1-=U)
    {
        C2<TF> q[U]; TF w3[U];
#pragma unroll
        for (int n=0; n<U; ++n) if (k1-n >= 0) { q[n] = p[col + (size_t)(k1-n)*ncol]; w3[n] = work3d[col + (size_t)(k1-n+1)*ncol]; }
#pragma unroll
        for (int n=0; n<U; ++n)
            if (k1-n >= 0)
            {
                C2<TF> r = q[n];
                r.x -= w3[n]*pp.x; r.y -= w3[n]*pp.y;
                q[n] = r; pp = r;
            }
#pragma unroll
        for (int n=0; n<U; ++n) if (k1-n >= 0) p[col + (size_t)(k1-n)*ncol] = q[n];
    }
}

static int tdma_factor(mhh_pres_plan* P)
{
    dim3 grid((unsigned)(((size_t)P->nxp*P->jtot + 63)/64));
    if (P->dtype == MHH_F64)
        hipLaunchKernelGGL(tdma_factor_kernel<double>, grid, dim3(64), 0, 0, (double*)P->work, cp<double>(P->bmati), cp<double>(P->bmatj),
                           cp<double>(P->a), cp<double>(P->c), cp<double>(P->dz), cp<double>(P->rhoref), P->nxh, P->nxp, P->jtot, P->ktot);
    else
        hipLaunchKernelGGL(tdma_factor_kernel<float>, grid, dim3(64), 0, 0, (float*)P->work, cp<float>(P->bmati), cp<float>(P->bmatj),
                           cp<float>(P->a), cp<float>(P->c), cp<float>(P->dz), cp<float>(P->rhoref), P->nxh, P->nxp, P->jtot, P->ktot);
    hipError_t h = hipGetLastError(); if (h == hipSuccess) h = hipStreamSynchronize(0);
    if (h != hipSuccess) { set_error("tdma_factor: %s", hipGetErrorString(h)); return MHH_EHIP; }
    return MHH_OK;
}

// =======================================================================================================
// spectral solve, pres_4: 7-band LU without pivoting on kmax+4 unknowns per column (src/pres_4.cxx:358-470, hdma :574-730)
// The band matrix depends on the grid and (kx,ky) only, so its LU factors are computed once at plan creation
// (hdma_factor_kernel, the reference's elimination order) and kept: band n (0..6) at W[(n*(kmax+4) + k)*ncol + col].
// Every solve is then the two substitution sweeps (hdma_solve_kernel), in place on the spectral array.
// =======================================================================================================
// one column: Wc = the column's element of band 0, row 0; row r of band n at Wc[n*bstride + r*ncol]
template<class TF>
__device__ __forceinline__ void hdma_factor_column(TF* __restrict__ Wc, size_t ncol, size_t bstride, TF bi, TF bj, bool mean,
                                                   const TF* __restrict__ M1, const TF* __restrict__ M2, const TF* __restrict__ M3, const TF* __restrict__ M4,
                                                   const TF* __restrict__ M5, const TF* __restrict__ M6, const TF* __restrict__ M7, int kmax)
{
    TF* __restrict__ m1 = Wc;             TF* __restrict__ m2 = Wc + bstride;   TF* __restrict__ m3 = Wc + 2*bstride; TF* __restrict__ m4 = Wc + 3*bstride;
    TF* __restrict__ m5 = Wc + 4*bstride; TF* __restrict__ m6 = Wc + 5*bstride; TF* __restrict__ m7 = Wc + 6*bstride;
#define A(arr, k) arr[(size_t)(k)*ncol]
    // fill (rows 0,1: bottom bc; 2..kmax+1: interior; kmax+2, kmax+3: top bc)
    A(m1,0)=0; A(m2,0)=0; A(m3,0)=0; A(m4,0)=1; A(m5,0)=0;  A(m6,0)=0; A(m7,0)=-1;
    A(m1,1)=0; A(m2,1)=0; A(m3,1)=0; A(m4,1)=1; A(m5,1)=-1; A(m6,1)=0; A(m7,1)=0;
    for (int k=0; k<kmax; ++k)
    {
        A(m1,k+2)=M1[k]; A(m2,k+2)=M2[k]; A(m3,k+2)=M3[k]; A(m4,k+2)=M4[k] + bi + bj;
        A(m5,k+2)=M5[k]; A(m6,k+2)=M6[k]; A(m7,k+2)=M7[k];
    }
    const int t = kmax+2;
    if (mean) { A(m1,t)=TF(0.);    A(m2,t)=TF(-1/3.); A(m3,t)=TF(2.);  A(m4,t)=TF(1.);
                A(m1,t+1)=TF(-2.); A(m2,t+1)=TF(9.);  A(m3,t+1)=TF(0.); A(m4,t+1)=TF(1.); }
    else      { A(m1,t)=TF(0.);    A(m2,t)=TF(0.);    A(m3,t)=TF(-1.); A(m4,t)=TF(1.);
                A(m1,t+1)=TF(-1.); A(m2,t+1)=TF(0.);  A(m3,t+1)=TF(0.); A(m4,t+1)=TF(1.); }
    A(m5,t)=0; A(m6,t)=0; A(m7,t)=0;
    A(m5,t+1)=0; A(m6,t+1)=0; A(m7,t+1)=0;
    // LU
    int k = 0;
    A(m1,k)=1; A(m2,k)=1; A(m3,k)=TF(1.)/A(m4,k); A(m4,k)=1; A(m5,k)=A(m5,k)*A(m3,k); A(m6,k)=A(m6,k)*A(m3,k); A(m7,k)=A(m7,k)*A(m3,k);
    k = 1;
    A(m1,k)=1; A(m2,k)=1; A(m3,k)=A(m3,k)/A(m4,k-1);
    A(m4,k)=A(m4,k)-A(m3,k)*A(m5,k-1); A(m5,k)=A(m5,k)-A(m3,k)*A(m6,k-1); A(m6,k)=A(m6,k)-A(m3,k)*A(m7,k-1);
    k = 2;
    A(m1,k)=1; A(m2,k)=A(m2,k)/A(m4,k-2);
    A(m3,k)=( A(m3,k) - A(m2,k)*A(m5,k-2) ) / A(m4,k-1);
    A(m4,k)=A(m4,k) - A(m3,k)*A(m5,k-1) - A(m2,k)*A(m6,k-2);
    A(m5,k)=A(m5,k) - A(m3,k)*A(m6,k-1) - A(m2,k)*A(m7,k-2);
    A(m6,k)=A(m6,k) - A(m3,k)*A(m7,k-1);
    for (k=3; k<kmax+4; ++k)
    {
        if (k == kmax+2) A(m7,kmax+1) = TF(1.);
        A(m1,k)=( A(m1,k) ) / A(m4,k-3);
        A(m2,k)=( A(m2,k) - A(m1,k)*A(m5,k-3) ) / A(m4,k-2);
        A(m3,k)=( A(m3,k) - A(m2,k)*A(m5,k-2) - A(m1,k)*A(m6,k-3) ) / A(m4,k-1);
        A(m4,k)=  A(m4,k) - A(m3,k)*A(m5,k-1) - A(m2,k)*A(m6,k-2) - A(m1,k)*A(m7,k-3);
        if (k < kmax+3) A(m5,k)= A(m5,k) - A(m3,k)*A(m6,k-1) - A(m2,k)*A(m7,k-2);
        if (k < kmax+2) A(m6,k)= A(m6,k) - A(m3,k)*A(m7,k-1);
        if (k == kmax+2) { A(m6,k)=TF(1.); A(m7,k)=TF(1.); }
        if (k == kmax+3) { A(m5,k)=TF(1.); A(m6,k)=TF(1.); A(m7,k)=TF(1.); }
    }
#undef A
}
template<class TF>
__global__ void __launch_bounds__(64) hdma_factor_kernel(TF* __restrict__ W,
                                                  const TF* __restrict__ bmati, const TF* __restrict__ bmatj,
                                                  const TF* __restrict__ M1, const TF* __restrict__ M2, const TF* __restrict__ M3, const TF* __restrict__ M4,
                                                  const TF* __restrict__ M5, const TF* __restrict__ M6, const TF* __restrict__ M7,
                                                  int nxh, int nxp, int jtot, int kmax)
{
    const int kx = blockIdx.x*64 + threadIdx.x, ky = blockIdx.y;
    if (kx >= nxh) return;
    const size_t ncol = (size_t)nxp*jtot, col = kx + (size_t)ky*nxp;
    hdma_factor_column(W + col, ncol, (size_t)(kmax+4)*ncol, bmati[kx], bmatj[ky], kx == 0 && ky == 0, M1, M2, M3, M4, M5, M6, M7, kmax);
}
// the same factors in the layout of the LDS form's y stage (pres_lds4.h, Pres4LdsSolve::F): column = (block row, ky), block row ncolx
// = the second modes of the threads that carry two; band 3 (the pivots) as reciprocals
template<class TF>
__global__ void __launch_bounds__(64) pres4_lds_factor_kernel(TF* __restrict__ F, const TF* __restrict__ bmati, const TF* __restrict__ bmatj,
                                                  const TF* __restrict__ M1, const TF* __restrict__ M2, const TF* __restrict__ M3, const TF* __restrict__ M4,
                                                  const TF* __restrict__ M5, const TF* __restrict__ M6, const TF* __restrict__ M7,
                                                  int ncolx, int jtot, int kmax)
{
    const int ky = blockIdx.x*64 + threadIdx.x, row = blockIdx.y;          // rows 0 .. ncolx
    if (ky >= jtot) return;
    const int kxa = (row == ncolx) ? ncolx : lds_fft::lds_mode_kx(row, ky, jtot, ncolx);
    const size_t wlev = (size_t)(ncolx + 1)*jtot, bstride = (size_t)(kmax+4)*wlev;
    TF* Fc = F + (size_t)row*jtot + ky;
    // (thread ky > N/2 of block 0 solves mode (itot/2, N - ky); bmatj is even in ky)
    hdma_factor_column(Fc, wlev, bstride, bmati[kxa], bmatj[ky], row == 0 && ky == 0, M1, M2, M3, M4, M5, M6, M7, kmax);
    for (int r=0; r<kmax+4; ++r) Fc[3*bstride + (size_t)r*wlev] = TF(1.) / Fc[3*bstride + (size_t)r*wlev];
}

static int hdma_factor(mhh_pres_plan* P)
{
    dim3 grid((P->nxh + 63)/64, P->jtot);
    if (P->dtype == MHH_F64)
        hipLaunchKernelGGL(hdma_factor_kernel<double>, grid, dim3(64), 0, 0, (double*)P->work, cp<double>(P->bmati), cp<double>(P->bmatj),
                           cp<double>(P->m[0]), cp<double>(P->m[1]), cp<double>(P->m[2]), cp<double>(P->m[3]), cp<double>(P->m[4]), cp<double>(P->m[5]), cp<double>(P->m[6]),
                           P->nxh, P->nxp, P->jtot, P->ktot);
    else
        hipLaunchKernelGGL(hdma_factor_kernel<float>, grid, dim3(64), 0, 0, (float*)P->work, cp<float>(P->bmati), cp<float>(P->bmatj),
                           cp<float>(P->m[0]), cp<float>(P->m[1]), cp<float>(P->m[2]), cp<float>(P->m[3]), cp<float>(P->m[4]), cp<float>(P->m[5]), cp<float>(P->m[6]),
                           P->nxh, P->nxp, P->jtot, P->ktot);
    hipError_t h = hipGetLastError(); if (h == hipSuccess) h = hipStreamSynchronize(0);
    if (h != hipSuccess) { set_error("hdma_factor: %s", hipGetErrorString(h)); return MHH_EHIP; }
    return MHH_OK;
}

// One thread per (column, re|im): the substitutions act on the two components independently. Unknown row r of the
// reference's kmax+4 system is spectral level r-2; the two boundary rows on either side have a zero right-hand side
// and live in registers only, so the sweeps run in place on p without a scratch copy of the rhs.
template<class TF>
__global__ void __launch_bounds__(128) hdma_solve_kernel(TF* __restrict__ p, const TF* __restrict__ W, size_t ncol, int nxh, int nxp, int kmax)
{
    const size_t t = (size_t)blockIdx.x*128 + threadIdx.x;
    if (t >= 2*ncol) return;
    const size_t col = t >> 1;
    if ((int)(col % nxp) >= nxh) return;                         // padding of the row pitch
    const int n = kmax+4;
    const TF* __restrict__ m1 = W + 0*(size_t)n*ncol + col; const TF* __restrict__ m2 = W + 1*(size_t)n*ncol + col;
    const TF* __restrict__ m3 = W + 2*(size_t)n*ncol + col; const TF* __restrict__ m4 = W + 3*(size_t)n*ncol + col;
    const TF* __restrict__ m5 = W + 4*(size_t)n*ncol + col; const TF* __restrict__ m6 = W + 5*(size_t)n*ncol + col;
    const TF* __restrict__ m7 = W + 6*(size_t)n*ncol + col;
    TF* __restrict__ q = p + t;                                  // level l of this component at q[l*2*ncol]
    const size_t ls = 2*ncol;
#define A(arr, k) arr[(size_t)(k)*ncol]
#define Q(r) q[(size_t)((r)-2)*ls]
    // L y = q  (rows 0 and 1 carry a zero rhs)
    TF a3 = TF(0.)*A(m3,0);
    TF a2 = TF(0.) - a3*A(m3,1);
    TF a1 = Q(2) - a2*A(m3,2) - a3*A(m2,2);
    Q(2) = a1;
    #pragma unroll 4
    for (int k=3; k<kmax+2; ++k)
    {
        const TF v = Q(k) - a1*A(m3,k) - a2*A(m2,k) - a3*A(m1,k);
        Q(k) = v;
        a3 = a2; a2 = a1; a1 = v;
    }
    TF yt0, yt1;                                                 // rows kmax+2, kmax+3
    { int k = kmax+2; yt0 = TF(0.) - a1*A(m3,k) - a2*A(m2,k) - a3*A(m1,k); a3 = a2; a2 = a1; a1 = yt0;
      k = kmax+3;     yt1 = TF(0.) - a1*A(m3,k) - a2*A(m2,k) - a3*A(m1,k); }
    // U x = y
    int k = kmax+3;
    TF b3 = yt1 / A(m4,k);
    TF b2 = ( yt0 - b3*A(m5,k-1) ) / A(m4,k-1);
    TF b1 = ( Q(k-2) - b2*A(m5,k-2) - b3*A(m6,k-2) ) / A(m4,k-2);
    Q(k-2) = b1;
    #pragma unroll 4
    for (k=kmax; k>=2; --k)
    {
        const TF v = ( Q(k) - b1*A(m5,k) - b2*A(m6,k) - b3*A(m7,k) ) / A(m4,k);
        Q(k) = v;
        b3 = b2; b2 = b1; b1 = v;
    }
#undef Q
#undef A
}

// =======================================================================================================
// unpack (src/pres_2.cxx:333-362, src/pres_4.cxx:481-528): normalise, ghosted layout, vertical ghost rows,
// periodic halo -- one kernel over icells x jcells x (kmax + vertical ghosts), reading with wrapped indices.
// =======================================================================================================
// Block b of a launch runs on XCD b % 8 (round-robin dispatch). chunk_of_block deals the chunks of a plane so that every
// XCD owns ONE contiguous run of them: the row j-1 a cell reads then sits in the L2 of the XCD that read it as row j
// (unpack+output at 512^3: HBM fetch 6.8 -> see profiles, r1j).
__device__ __forceinline__ int chunk_of_block(int b, int nb)
{
    const int per = nb >> 3, rem = nb & 7, xcd = b & 7, idx = b >> 3;
    return xcd*per + (xcd < rem ? xcd : rem) + idx;
}

// Threads run over the flattened (i, j) plane of the ghosted layout (rows of icells are contiguous, so a wave still stores
// one contiguous run): no thread of a block is idle whatever icells is.
template<bool POW2, class TF>
__global__ void __launch_bounds__(256) unpack_kernel(TF* __restrict__ p, const TF* __restrict__ packed, int order,
                                                     int itot, int jtot, int kmax, int igc, int jgc, int kgc, int icells, int jcells)
{
    const int ijc = chunk_of_block(blockIdx.x, gridDim.x)*256 + threadIdx.x;
    const int kz = blockIdx.z;                 // 0 .. kmax-1 + nghost rows
    if (ijc >= icells*jcells) return;
    const int j = ijc / icells, i = ijc - j*icells;
    const TF ri = TF(1)/TF(itot), rj = TF(1)/TF(jtot);
    // destination level and source level
    int kd, ks;
    if (kz < kmax) { kd = kz + kgc; ks = kz; }
    else if (order == 2) { kd = kgc - 1; ks = 0; }                        // p[kstart-1] = p[kstart]
    else
    {
        const int gidx = kz - kmax;                                         // 0..3
        if (gidx == 0)      { kd = kgc - 1;      ks = 0; }                 // p[kstart-1] = p[kstart]
        else if (gidx == 1) { kd = kgc - 2;      ks = 1; }                 // p[kstart-2] = p[kstart+1]
        else if (gidx == 2) { kd = kgc + kmax;   ks = kmax-1; }            // p[kend]   = p[kend-1]
        else                { kd = kgc + kmax+1; ks = kmax-2; }            // p[kend+1] = p[kend-2]
    }
    int is = (i - igc) % itot; if (is < 0) is += itot;
    int js = (j - jgc) % jtot; if (js < 0) js += jtot;
    const TF val = fft_norm<POW2>(packed[(size_t)is + (size_t)js*itot + (size_t)ks*itot*jtot], itot, jtot, ri, rj);
    p[(size_t)ijc + (size_t)kd*icells*jcells] = val;
}

// the k-sweep per (kx, ky) column between the two transforms: Thomas (pres_2) or the factored 7-band substitution (pres_4)
static int pres_column_solve(mhh_pres_plan* P, const mhh_grid* g, hipStream_t st)
{
    dim3 grid((P->nxh + 63)/64, P->jtot);
    if (P->order == 2)
    {
        grid = dim3((unsigned)(((size_t)P->nxp*P->jtot + 63)/64));
        if (g->dtype == MHH_F64)
            hipLaunchKernelGGL(tdma_kernel<double>, grid, dim3(64), 0, st, (C2<double>*)P->spec, cp<double>(P->work), cp<double>(P->bmati), cp<double>(P->bmatj),
                               cp<double>(P->a), cp<double>(P->c), cp<double>(P->dz), cp<double>(P->rhoref), P->nxh, P->nxp, P->jtot, P->ktot);
        else
            hipLaunchKernelGGL(tdma_kernel<float>, grid, dim3(64), 0, st, (C2<float>*)P->spec, cp<float>(P->work), cp<float>(P->bmati), cp<float>(P->bmatj),
                               cp<float>(P->a), cp<float>(P->c), cp<float>(P->dz), cp<float>(P->rhoref), P->nxh, P->nxp, P->jtot, P->ktot);
    }
    else
    {
        const size_t ncol = (size_t)P->nxp*P->jtot;
        dim3 sg((unsigned)((2*ncol + 127)/128));
        if (g->dtype == MHH_F64) hipLaunchKernelGGL(hdma_solve_kernel<double>, sg, dim3(128), 0, st, (double*)P->spec, cp<double>(P->work), ncol, P->nxh, P->nxp, P->ktot);
        else                     hipLaunchKernelGGL(hdma_solve_kernel<float>,  sg, dim3(128), 0, st, (float*)P->spec,  cp<float>(P->work),  ncol, P->nxh, P->nxp, P->ktot);
    }
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}

// unpack and Pres_2::output in one pass (src/pres_2.cxx:333-387): every thread normalises and stores its own p (interior,
// halo or bottom ghost row, as unpack_kernel) and, on interior cells, also the three neighbours the pressure gradient
// needs -- re-normalised from the packed solution with the same two divisions, so that the values are the stored ones --
// and corrects ut, vt, wt. One read of p less, one kernel less than unpack + output.
template<bool POW2, class TF>
__global__ void __launch_bounds__(256) unpack_out2_kernel(TF* __restrict__ p, const TF* __restrict__ packed,
                                                          TF* __restrict__ ut, TF* __restrict__ vt, TF* __restrict__ wt, const TF* __restrict__ dzhi,
                                                          TF dxi, TF dyi, int itot, int jtot, int kmax, int igc, int jgc, int kgc, int icells, int jcells,
                                                          TF* __restrict__ u, TF* __restrict__ v, TF* __restrict__ w, TF cA, TF cB, TF rdt)
{
    // u != nullptr: the Runge-Kutta sub-step of u, v, w on the corrected tendencies (see pres_lds.h, PresLdsOut)
    const int ijc = chunk_of_block(blockIdx.x, gridDim.x)*256 + threadIdx.x;
    const int kz = blockIdx.z;                 // 0 .. kmax-1, kmax = the bottom ghost row
    if (ijc >= icells*jcells) return;
    const int j = ijc / icells, i = ijc - j*icells;
    const TF ri = TF(1)/TF(itot), rj = TF(1)/TF(jtot);
    const int kd = (kz < kmax) ? kz + kgc : kgc - 1;
    const int ks = (kz < kmax) ? kz : 0;
    int is = (i - igc) % itot; if (is < 0) is += itot;
    int js = (j - jgc) % jtot; if (js < 0) js += jtot;
    const size_t ij = (size_t)itot*jtot;
    const TF pc = fft_norm<POW2>(packed[(size_t)is + (size_t)js*itot + (size_t)ks*ij], itot, jtot, ri, rj);
    const size_t c = (size_t)ijc + (size_t)kd*icells*jcells;
    p[c] = pc;
    if (kz < kmax && i >= igc && i < igc + itot && j >= jgc && j < jgc + jtot)
    {
        const int iw = (is == 0) ? itot-1 : is-1, jsm = (js == 0) ? jtot-1 : js-1;
        const TF pw = fft_norm<POW2>(packed[(size_t)iw + (size_t)js *itot + (size_t)ks*ij], itot, jtot, ri, rj);
        const TF ps = fft_norm<POW2>(packed[(size_t)is + (size_t)jsm*itot + (size_t)ks*ij], itot, jtot, ri, rj);
        const TF pb = (ks == 0) ? pc : fft_norm<POW2>(packed[(size_t)is + (size_t)js*itot + (size_t)(ks-1)*ij], itot, jtot, ri, rj);   // p[kstart-1] = p[kstart]
        if (u == nullptr)
        {
            ut[c] -= (pc - pw) * dxi;
            vt[c] -= (pc - ps) * dyi;
            wt[c] -= (pc - pb) * dzhi[kd];
        }
        else
        {
            const TF nut = ut[c] - (pc - pw) * dxi, nvt = vt[c] - (pc - ps) * dyi, nwt = wt[c] - (pc - pb) * dzhi[kd];
            u[c] = u[c] + cB*rdt*nut; ut[c] = cA*nut;
            v[c] = v[c] + cB*rdt*nvt; vt[c] = cA*nvt;
            w[c] = w[c] + cB*rdt*nwt; wt[c] = cA*nwt;
        }
    }
}

// The same for pres_4 (src/pres_4.cxx:481-571): the four mirrored ghost levels of p are written like unpack_kernel does, and an
// interior cell gathers the 4-point gradient stencils in x, y and z straight from the packed solution -- wrapped in the
// horizontal, mirrored across the walls in the vertical (p[kstart-1] = p[kstart], p[kstart-2] = p[kstart+1], likewise on top).
template<bool POW2, class TF>
__global__ void __launch_bounds__(256) unpack_out4_kernel(TF* __restrict__ p, const TF* __restrict__ packed,
                                                          TF* __restrict__ ut, TF* __restrict__ vt, TF* __restrict__ wt, const TF* __restrict__ dzhi4,
                                                          TF dxi, TF dyi, int dim3, int itot, int jtot, int kmax, int igc, int jgc, int kgc, int icells, int jcells)
{
    const int ijc = chunk_of_block(blockIdx.x, gridDim.x)*256 + threadIdx.x;
    const int kz = blockIdx.z;                 // 0 .. kmax-1, then the four ghost levels
    if (ijc >= icells*jcells) return;
    const int j = ijc / icells, i = ijc - j*icells;
    const TF ri = TF(1)/TF(itot), rj = TF(1)/TF(jtot);
    int kd, ks;
    if (kz < kmax) { kd = kz + kgc; ks = kz; }
    else
    {
        const int gidx = kz - kmax;
        if (gidx == 0)      { kd = kgc - 1;      ks = 0; }
        else if (gidx == 1) { kd = kgc - 2;      ks = 1; }
        else if (gidx == 2) { kd = kgc + kmax;   ks = kmax-1; }
        else                { kd = kgc + kmax+1; ks = kmax-2; }
    }
    int is = (i - igc) % itot; if (is < 0) is += itot;
    int js = (j - jgc) % jtot; if (js < 0) js += jtot;
    const size_t ij = (size_t)itot*jtot;
    auto P = [&](int ii, int jw, int kk) -> TF                 // normalised solution at wrapped (ii, jw), mirrored level kk
    {
        int iw = ii % itot; if (iw < 0) iw += itot;
        int jv = jw % jtot; if (jv < 0) jv += jtot;
        const int kv = (kk < 0) ? -kk-1 : (kk >= kmax ? 2*kmax-1-kk : kk);
        return fft_norm<POW2>(packed[(size_t)iw + (size_t)jv*itot + (size_t)kv*ij], itot, jtot, ri, rj);
    };
    const TF pc = P(is, js, ks);
    const size_t c = (size_t)ijc + (size_t)kd*icells*jcells;
    p[c] = pc;
    if (kz < kmax && i >= igc && i < igc + itot && j >= jgc && j < jgc + jtot)
    {
        ut[c] -= cg4(P(is-2, js, ks), P(is-1, js, ks), pc, P(is+1, js, ks)) * dxi;
        if (dim3) vt[c] -= cg4(P(is, js-2, ks), P(is, js-1, ks), pc, P(is, js+1, ks)) * dyi;
        if (ks > 0) wt[c] -= cg4(P(is, js, ks-2), P(is, js, ks-1), pc, P(is, js, ks+1)) * dzhi4[kd];
    }
}

// forward transform, column solves, inverse transform: packed rhs -> packed (un-normalised) solution
static int pres_spectral(mhh_pres_plan* P, const mhh_grid* g, void* p_packed, hipStream_t st)
{
    MHH_FFT_TRY(rocfft_execution_info_set_stream(P->fwd_info, st));
    MHH_FFT_TRY(rocfft_execution_info_set_stream(P->bwd_info, st));
    void* in[1] = {p_packed}; void* out[1] = {P->spec};
    MHH_FFT_TRY(rocfft_execute(P->fwd, in, out, P->fwd_info));
    if (int e = pres_column_solve(P, g, st)) return e;
    void* in2[1] = {P->spec}; void* out2[1] = {p_packed};
    MHH_FFT_TRY(rocfft_execute(P->bwd, in2, out2, P->bwd_info));
    return MHH_OK;
}
MHH_API int mhh_pres_solve(mhh_pres_plan* P, const mhh_grid* g, const mhh_fields* f, void* p_packed, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(P && f && f->p, "null field");
    MHH_REQUIRE(P->dtype == g->dtype && P->itot == g->itot && P->jtot == g->jtot && P->ktot == g->ktot, "plan/grid mismatch");
    if (!p_packed) p_packed = P->packed;
    hipStream_t st = as_stream(stream);
    if (int e = pres_spectral(P, g, p_packed, st)) return e;
    const int nghost = (P->order == 2) ? 1 : 4;
    dim3 ug((g->icells*g->jcells + 255)/256, 1, g->kmax + nghost);
    const bool pow2 = is_pow2(g->itot) && is_pow2(g->jtot);
#define UNPACK(POW2, TF) hipLaunchKernelGGL((unpack_kernel<POW2, TF>), ug, dim3(256), 0, st, mp<TF>(f->p), cp<TF>(p_packed), P->order, g->itot, g->jtot, g->kmax, g->igc, g->jgc, g->kgc, g->icells, g->jcells)
    if (g->dtype == MHH_F64) { if (pow2) UNPACK(true, double); else UNPACK(false, double); }
    else                     { if (pow2) UNPACK(true, float);  else UNPACK(false, float); }
#undef UNPACK
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}

// =======================================================================================================
// output (src/pres_2.cxx:365-387, src/pres_4.cxx:533-571)
// =======================================================================================================
template<class TF>
struct PresOutOp
{
    GridDev<TF> g; int order; TF* __restrict__ ut; TF* __restrict__ vt; TF* __restrict__ wt; const TF* __restrict__ p;
    __device__ void operator()(int, int, int k, int c) const
    {
        const int jj = g.icells, kk = g.ijcells;
        if (order == 2)
        {
            ut[c] -= (p[c] - p[c-1 ]) * g.dxi_t;
            vt[c] -= (p[c] - p[c-jj]) * g.dyi_t;
            wt[c] -= (p[c] - p[c-kk]) * g.dzhi[k];
        }
        else
        {
            ut[c] -= cg4(p[c-2], p[c-1], p[c], p[c+1]) * g.dxi_d;
            if (g.dim3) vt[c] -= cg4(p[c-2*jj], p[c-jj], p[c], p[c+jj]) * g.dyi_d;
            if (k > g.kstart) wt[c] -= cg4(p[c-2*kk], p[c-kk], p[c], p[c+kk]) * g.dzhi4[k];
        }
    }
};
MHH_API int mhh_pres_output_order(const mhh_grid* g, int order, const mhh_fields* f, void* stream);
MHH_API int mhh_pres_output(mhh_pres_plan* P, const mhh_grid* g, const mhh_fields* f, void* stream)
{
    MHH_REQUIRE(P != nullptr, "plan");
    return mhh_pres_output_order(g, P->order, f, stream);
}
MHH_API int mhh_pres_output_order(const mhh_grid* g, int order, const mhh_fields* f, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(f && f->p && f->ut && f->vt && f->wt, "null field");
    MHH_REQUIRE(order == 2 || order == 4, "order");
#define CALL(TF) [&]{ PresOutOp<TF> op{make_grid<TF>(g), order, mp<TF>(f->ut), mp<TF>(f->vt), mp<TF>(f->wt), cp<TF>(f->p)}; \
                      return launch_interior(as_stream(stream), op.g, g->kstart, g->kend, op); }()
    return MHH_DISPATCH(g, CALL);
#undef CALL
}

// =======================================================================================================
// Fused form of Pres::exec. rocFFT lets a transform obtain its input through a load callback and deliver its output
// through a store callback (rocfft_execution_info_set_load_callback / _store_callback). Pres::input is such a producer
// (one value per packed cell, from the six ghosted fields) and the unpack such a consumer (normalise, ghosted layout,
// periodic halo and vertical ghost rows), so the packed right-hand side and the packed solution never exist in memory:
// two array writes and two array reads less than the staged form, and two kernels less. The transforms themselves and the
// arithmetic of both ends are the staged form's: same bits (tests: fused vs staged).
// =======================================================================================================
template<class TF> struct PresCb
{
    GridDev<TF> g; int order;
    const TF* u; const TF* v; const TF* w; const TF* ut; const TF* vt; const TF* wt; const TF* rhoref; const TF* rhorefh;
    TF dti2, dti4;
    TF* p;
};
template<class TF>
__device__ TF pres_load_cb(TF*, size_t offset, void* cbdata, void*)
{
    const PresCb<TF>& d = *static_cast<const PresCb<TF>*>(cbdata);
    const GridDev<TF>& g = d.g;
    const unsigned o = (unsigned)offset, ij = (unsigned)g.imax*(unsigned)g.jmax;
    const unsigned kz = o / ij, r = o - kz*ij, jz = r / (unsigned)g.imax, iz = r - jz*(unsigned)g.imax;
    const int k = (int)kz + g.kgc;
    const int c = ((int)iz + g.igc) + ((int)jz + g.jgc)*g.icells + k*g.ijcells;
    if (d.order == 2) return pres2_in(d.u, d.v, d.w, d.ut, d.vt, d.wt, c, g.icells, g.ijcells, g.dxi_t, g.dyi_t, d.dti2, d.rhoref[k], d.rhorefh[k], d.rhorefh[k+1], g.dzi[k]);
    return pres4_in(d.u, d.v, d.w, d.ut, d.vt, d.wt, c, g.icells, g.ijcells, g.dxi_d, g.dyi_d, d.dti4, g.dzi4[k], g.dim3);
}
template<class TF>
__device__ void pres_store_cb(TF*, size_t offset, TF value, void* cbdata, void*)
{
    const PresCb<TF>& d = *static_cast<const PresCb<TF>*>(cbdata);
    const GridDev<TF>& g = d.g;
    const int itot = g.imax, jtot = g.jmax, kmax = g.kend - g.kstart;
    const unsigned o = (unsigned)offset, ij = (unsigned)itot*(unsigned)jtot;
    const unsigned kz = o / ij, r = o - kz*ij, jz = r / (unsigned)itot, iz = r - jz*(unsigned)itot;
    const TF val = value / jtot / itot;                              // as unpack_kernel (src/fft.cxx scales 1/jtot then 1/itot)
    // destination levels: the interior one and the ghost rows that mirror it (src/pres_2.cxx:352-360, src/pres_4.cxx:508-525)
    int kd[2]; int nk = 1; kd[0] = (int)kz + g.kgc;
    if (d.order == 2) { if (kz == 0) kd[nk++] = g.kgc - 1; }
    else
    {
        if (kz == 0) kd[nk++] = g.kgc - 1;
        if (kz == 1) kd[nk++] = g.kgc - 2;
        if ((int)kz == kmax-1) kd[nk++] = g.kgc + kmax;
        if ((int)kz == kmax-2) kd[nk++] = g.kgc + kmax + 1;
    }
    for (int n=0; n<nk; ++n)
        for (int jd = ((int)jz + g.jgc) % jtot; jd < g.jcells; jd += jtot)       // every row / column congruent to this one: the periodic halo
            for (int id = ((int)iz + g.igc) % itot; id < g.icells; id += itot)
                d.p[(size_t)id + (size_t)jd*g.icells + (size_t)kd[n]*g.ijcells] = val;
}
// the record travels as a kernel argument: stream-ordered, and no host buffer has to outlive the call
template<class TF> __global__ void pres_cb_record(PresCb<TF>* dst, const PresCb<TF> src) { *dst = src; }
template<class TF> __global__ void pres_cb_addresses(void** out) { out[0] = (void*)&pres_load_cb<TF>; out[1] = (void*)&pres_store_cb<TF>; }

static int pres_cb_setup(mhh_pres_plan* P)
{
    // device addresses of the two callbacks, execution infos that carry them, the record they read
    void** dptr = nullptr; void* h[2] = {nullptr, nullptr};
    MHH_HIP_TRY(hipMalloc((void**)&dptr, 2*sizeof(void*)));
    if (P->dtype == MHH_F64) hipLaunchKernelGGL(pres_cb_addresses<double>, dim3(1), dim3(1), 0, 0, dptr);
    else                     hipLaunchKernelGGL(pres_cb_addresses<float>,  dim3(1), dim3(1), 0, 0, dptr);
    hipError_t he = hipMemcpy(h, dptr, 2*sizeof(void*), hipMemcpyDeviceToHost);
    (void)hipFree(dptr);
    if (he != hipSuccess || !h[0] || !h[1]) { set_error("pres callbacks: address query failed"); return MHH_EHIP; }
    MHH_HIP_TRY(hipMalloc(&P->cb_data, sizeof(PresCb<double>)));
    MHH_FFT_TRY(rocfft_execution_info_create(&P->fwd_info_cb));
    MHH_FFT_TRY(rocfft_execution_info_create(&P->bwd_info_cb));
    size_t wf = 0, wb = 0;
    MHH_FFT_TRY(rocfft_plan_get_work_buffer_size(P->fwd, &wf));
    MHH_FFT_TRY(rocfft_plan_get_work_buffer_size(P->bwd, &wb));
    if (wf) MHH_FFT_TRY(rocfft_execution_info_set_work_buffer(P->fwd_info_cb, P->fwd_wb, wf));
    if (wb) MHH_FFT_TRY(rocfft_execution_info_set_work_buffer(P->bwd_info_cb, P->bwd_wb, wb));
    void* lfn[1] = {h[0]}; void* sfn[1] = {h[1]}; void* dat[1] = {P->cb_data};
    if (rocfft_execution_info_set_load_callback(P->fwd_info_cb, lfn, dat, 0) != rocfft_status_success) return MHH_OK;   // not supported: stay staged
    if (rocfft_execution_info_set_store_callback(P->bwd_info_cb, sfn, dat, 0) != rocfft_status_success) return MHH_OK;
    P->cb_ready = true;
    return MHH_OK;
}

template<class TF>
static int pres_exec_fused(mhh_pres_plan* P, const mhh_grid* g, const mhh_fields* f, double dt, hipStream_t st)
{
    PresCb<TF> h;
    h.g = make_grid<TF>(g); h.order = P->order;
    h.u = cp<TF>(f->u); h.v = cp<TF>(f->v); h.w = cp<TF>(f->w); h.ut = cp<TF>(f->ut); h.vt = cp<TF>(f->vt); h.wt = cp<TF>(f->wt);
    h.rhoref = cp<TF>(f->rhoref); h.rhorefh = cp<TF>(f->rhorefh); h.dti2 = TF(1.)/TF(dt); h.dti4 = TF(1./TF(dt)); h.p = mp<TF>(f->p);
    hipLaunchKernelGGL(pres_cb_record<TF>, dim3(1), dim3(1), 0, st, static_cast<PresCb<TF>*>(P->cb_data), h);
    MHH_LAUNCH_CHECK();
    MHH_FFT_TRY(rocfft_execution_info_set_stream(P->fwd_info_cb, st));
    MHH_FFT_TRY(rocfft_execution_info_set_stream(P->bwd_info_cb, st));
    void* in[1] = {P->packed}; void* out[1] = {P->spec};          // `packed` is never read or written: the callbacks stand in for it
    MHH_FFT_TRY(rocfft_execute(P->fwd, in, out, P->fwd_info_cb));
    if (int e = pres_column_solve(P, g, st)) return e;
    void* in2[1] = {P->spec}; void* out2[1] = {P->packed};
    MHH_FFT_TRY(rocfft_execute(P->bwd, in2, out2, P->bwd_info_cb));
    return MHH_OK;
}

// =======================================================================================================
// Pres_2::exec with the transforms in LDS (pres_lds.h): input + x transform | y transforms around the Thomas sweeps |
// x transform + p + output. Power-of-two itot (16 .. 1024) and jtot (8 .. 1024) whose rows fit the LDS; everything else
// takes the staged form above.
// =======================================================================================================
#ifndef MHH_PRES_LDS_RG
#define MHH_PRES_LDS_RG 2          // rows whose loads a thread of the x-stage kernels keeps in flight together
#endif
static constexpr int LDS_RG = MHH_PRES_LDS_RG;
// Every power-of-two row length has an instantiation of its own with the transform's strides as compile-time constants and a
// block of exactly that many threads (round 2 had one per precision, the benchmark's; every other shape took run-time-size kernels
// whose fp64 forms spilled at the 128 registers of a 1024-thread block). X rows: itot = 128 ... 1024 (NX = log2(itot/2) = 6 ... 9),
// y rows: jtot = 64 ... 1024 (NY = 6 ... 10); shorter rows (tests, toy grids) run the run-time-size kernels in small blocks, where
// registers are plentiful. Launch bounds keep at least 1024 threads per CU resident. An instantiation that needs scratch on this
// compiler is not used (hipFuncGetAttributes at plan creation): the plan then has no LDS form and takes the staged one.
#define MHH_FOR_NX(M) M(6) M(7) M(8) M(9)
#define MHH_FOR_NY(M) M(6) M(7) M(8) M(9) M(10)
#define MHH_FOR_NX_T(M, T) M(T, 6) M(T, 7) M(T, 8) M(T, 9)
#define MHH_FOR_NY_T(M, T) M(T, 6) M(T, 7) M(T, 8) M(T, 9) M(T, 10)
// (fp64 rows of 1024 along y: the unrolled passes do not fit the 128 registers of a 1024-thread block -- 88 bytes of scratch -- so
// that size has no instantiation and such grids take the staged form)
template<class TF, int NY> static constexpr bool lds_has_ny() { return sizeof(TF) == 4 || NY < 10; }
static constexpr int LDS_XS = 64, LDS_YS = 32;            // block sizes of the run-time-size forms (itot <= 64, jtot <= 32)
static int ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }
// rows of the transforms + the twiddle table; the 9-row kernel (stage 3) also keeps p of the level below there (8 rows of itot reals)
// (pres_4: its 8-row kernel keeps a carried level there, its 11-row kernel nothing)
static size_t lds_bytes_x(const mhh_pres_plan* P, int rows) { return ((size_t)rows*(P->itot/2 + 2) + P->itot) * 2*P->esz + ((rows == 9 || (rows == 8 && P->order == 4)) ? (size_t)8*P->itot*P->esz : 0); }
static size_t lds_bytes_y(const mhh_pres_plan* P)           { return ((size_t)8*P->jtot + P->jtot) * 2*P->esz; }
template<class TF>
static lds_fft::PresLdsSolve<TF> lds_solve_args(const mhh_pres_plan* P)
{
    return lds_fft::PresLdsSolve<TF>{static_cast<C2<TF>*>(P->spec), cp<TF>(P->w3l), cp<TF>(P->bmati), cp<TF>(P->bmatj), cp<TF>(P->a), cp<TF>(P->c),
                                     cp<TF>(P->dz), cp<TF>(P->rhoref), static_cast<const C2<TF>*>(P->ty), P->itot/2, P->jtot, ilog2(P->jtot), P->ktot};
}
// the kernel of a stage for a row length: NX / NY = 0 selects the run-time-size form
template<class TF, int NX> static const void* lds_kernel_in()  { return reinterpret_cast<const void*>(&lds_fft::pres_in_fftx_kernel<TF, LDS_RG, (NX ? (2 << NX) : LDS_XS), NX>); }
template<class TF, int NX> static const void* lds_kernel_out() { return reinterpret_cast<const void*>(&lds_fft::pres_ifftx_out_kernel<TF, LDS_RG, (NX ? (2 << NX) : LDS_XS), NX>); }
template<class TF, int NY> static const void* lds_kernel_y()   { return reinterpret_cast<const void*>(&lds_fft::pres_ysolve_kernel<TF, (NY ? (1 << NY) : LDS_YS), NY>); }
template<class TF, int NX> static const void* lds4_kernel_in()  { return reinterpret_cast<const void*>(&lds_fft::pres4_in_fftx_kernel<TF, (NX ? (2 << NX) : LDS_XS), NX>); }
template<class TF, int NX> static const void* lds4_kernel_out() { return reinterpret_cast<const void*>(&lds_fft::pres4_ifftx_out_kernel<TF, (NX ? (2 << NX) : LDS_XS), NX>); }
template<class TF, int NY> static const void* lds4_kernel_y()   { return reinterpret_cast<const void*>(&lds_fft::pres4_ysolve_kernel<TF, (NY ? (1 << NY) : LDS_YS), NY>); }
template<class TF, int NY, int PH> static const void* lds_kernel_ytw() { return reinterpret_cast<const void*>(&lds_fft::pres_ysolve_tw_kernel<TF, (NY ? (1 << NY) : LDS_YS), NY, PH>); }
// once per process and kernel: the dynamic-LDS ceiling at the device's maximum (the attribute belongs to the FUNCTION, not to a
// plan: set per plan to that plan's bytes, a small plan created after a large one lowered the ceiling under the large one), and
// whether the instantiation needs scratch
static int lds_kernel_ready(const void* kernel, bool& usable)
{
    static std::map<const void*, bool> seen;
    static std::mutex mtx;
    std::lock_guard<std::mutex> lock(mtx);
    auto it = seen.find(kernel);
    if (it == seen.end())
    {
        MHH_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
        hipFuncAttributes fa;
        MHH_HIP_TRY(hipFuncGetAttributes(&fa, kernel));
        it = seen.emplace(kernel, fa.localSizeBytes == 0).first;
    }
    usable = it->second;
    return MHH_OK;
}
template<class TF>
static int pres_lds_kernels_ready(const mhh_pres_plan* P, bool& usable)
{
    const int nx = ilog2(P->itot/2), ny = ilog2(P->jtot);
    const void* k[3] = {nullptr, nullptr, nullptr};
    if (P->order == 2)
    {
        if (P->itot <= LDS_XS) { k[0] = lds_kernel_in<TF, 0>(); k[1] = lds_kernel_out<TF, 0>(); }
        if (P->jtot <= LDS_YS) k[2] = lds_kernel_y<TF, 0>();
#define M(N) if (nx == N) { k[0] = lds_kernel_in<TF, N>(); k[1] = lds_kernel_out<TF, N>(); }
        MHH_FOR_NX(M)
#undef M
#define M(N) if (ny == N) { if constexpr (lds_has_ny<TF, N>()) k[2] = lds_kernel_y<TF, N>(); }
        MHH_FOR_NY(M)
#undef M
    }
    else
    {
        if (P->itot <= LDS_XS) { k[0] = lds4_kernel_in<TF, 0>(); k[1] = lds4_kernel_out<TF, 0>(); }
        if (P->jtot <= LDS_YS) k[2] = lds4_kernel_y<TF, 0>();
#define M(N) if (nx == N) { k[0] = lds4_kernel_in<TF, N>(); k[1] = lds4_kernel_out<TF, N>(); }
        MHH_FOR_NX(M)
#undef M
#define M(N) if (ny == N) { if constexpr (lds_has_ny<TF, N>()) k[2] = lds4_kernel_y<TF, N>(); }
        MHH_FOR_NY(M)
#undef M
    }
    usable = true;
    for (int n=0; n<3; ++n)
    {
        if (!k[n]) { usable = false; return MHH_OK; }
        bool u = false;
        if (int e = lds_kernel_ready(k[n], u)) return e;
        usable = usable && u;
    }
    return MHH_OK;
}
template<class TF>
static int pres_lds_setup_t(mhh_pres_plan* P)
{
    bool usable = false;
    if (int e = pres_lds_kernels_ready<TF>(P, usable)) return e;
    if (!usable) return MHH_OK;
    const double pi = std::acos(-1.);
    for (int d=0; d<2; ++d)
    {
        const int n = d ? P->jtot : P->itot;
        std::vector<TF> t(2*(size_t)n);
        for (int m=0; m<n; ++m) { t[2*m] = (TF)std::cos(2.*pi*m/n); t[2*m+1] = (TF)(-std::sin(2.*pi*m/n)); }
        // the quarter points exactly
        t[0] = 1; t[1] = 0; t[n] = -1; t[n+1] = 0;
        if (n >= 4) { t[n/2] = 0; t[n/2+1] = -1; t[3*n/2] = 0; t[3*n/2+1] = 1; }
        if (int e = upload(d ? &P->ty : &P->tx, t)) return e;
    }
    if (P->order == 2)
    {
        MHH_HIP_TRY(hipMalloc(&P->w3l, (size_t)(P->itot/2 + 1)*P->jtot*P->ktot*sizeof(TF)));
        hipLaunchKernelGGL(lds_fft::pres_lds_factor_kernel<TF>, dim3((P->jtot + 63)/64, P->itot/2 + 1), dim3(64), 0, 0, static_cast<TF*>(P->w3l), lds_solve_args<TF>(P));
        // the two-blocks-per-column form of the y stage: its kernels (no scratch), the factors of the top-down half, the buffer where the halves meet
        // (only where pres_y_twisted() takes it by itself, or when the switch is set at plan creation: its tables are another spectral array)
        if (P->ktot >= 16 && (P->itot <= 256 || getenv("MHH_PRES_Y_TWISTED")))
        {
            const int ny = ilog2(P->jtot);
            const void* k2[2] = {nullptr, nullptr};
            if (P->jtot <= LDS_YS) { k2[0] = lds_kernel_ytw<TF, 0, 1>(); k2[1] = lds_kernel_ytw<TF, 0, 2>(); }
#define M(N) if (ny == N) { if constexpr (lds_has_ny<TF, N>()) { k2[0] = lds_kernel_ytw<TF, N, 1>(); k2[1] = lds_kernel_ytw<TF, N, 2>(); } }
            MHH_FOR_NY(M)
#undef M
            bool ok2 = (k2[0] != nullptr);
            for (int n=0; n<2 && ok2; ++n) { bool u = false; if (int e = lds_kernel_ready(k2[n], u)) return e; ok2 = u; }
            if (ok2)
            {
                P->ksplit = ((P->ktot/2 + 7)/8)*8;
                MHH_HIP_TRY(hipMalloc(&P->a3l, (size_t)(P->itot/2 + 1)*P->jtot*P->ktot*sizeof(TF)));
                MHH_HIP_TRY(hipMalloc(&P->itw, (size_t)(P->itot/2)*2*P->jtot*2*sizeof(TF)));
                const lds_fft::PresLdsSolveTw<TF> ta{lds_solve_args<TF>(P), cp<TF>(P->a3l), static_cast<C2<TF>*>(P->itw), P->ksplit};
                hipLaunchKernelGGL(lds_fft::pres_lds_factor_tw_kernel<TF>, dim3((P->jtot + 63)/64, P->itot/2 + 1), dim3(64), 0, 0, static_cast<TF*>(P->a3l), ta);
                P->tw_ok = true;
            }
        }
    }
    else
    {   // the 7 factored bands once more, in the column order of this form's y stage (the staged form keeps its own in P->work)
        MHH_HIP_TRY(hipMalloc(&P->w3l, (size_t)7*(P->ktot + 4)*(P->itot/2 + 1)*P->jtot*sizeof(TF)));
        hipLaunchKernelGGL(pres4_lds_factor_kernel<TF>, dim3((P->jtot + 63)/64, P->itot/2 + 1), dim3(64), 0, 0, static_cast<TF*>(P->w3l), cp<TF>(P->bmati), cp<TF>(P->bmatj),
                           cp<TF>(P->m[0]), cp<TF>(P->m[1]), cp<TF>(P->m[2]), cp<TF>(P->m[3]), cp<TF>(P->m[4]), cp<TF>(P->m[5]), cp<TF>(P->m[6]), P->itot/2, P->jtot, P->ktot);
    }
    MHH_LAUNCH_CHECK();
    MHH_HIP_TRY(hipStreamSynchronize(0));
    P->lds_ok = true;
    return MHH_OK;
}
static int pres_lds_setup(mhh_pres_plan* P, const mhh_grid* g)
{
    const size_t lds_max = 160*1024;
    if (!(is_pow2(P->itot) && P->itot >= 16 && P->itot <= 1024 && is_pow2(P->jtot) && P->jtot >= 8 && P->jtot <= 1024)) return MHH_OK;
    if (lds_bytes_x(P, P->order == 2 ? 9 : 11) > lds_max || lds_bytes_x(P, 8) > lds_max || lds_bytes_y(P) > lds_max) return MHH_OK;
    if (g->igc > P->itot || g->jgc > P->jtot) return MHH_OK;
    if ((long long)g->icells*g->jcells*g->kcells >= (1ll << 31)) return MHH_OK;        // the x-stage kernels index cells with 32 bits
    return (P->dtype == MHH_F64) ? pres_lds_setup_t<double>(P) : pres_lds_setup_t<float>(P);
}
// the y stage with two blocks per column (pres_lds.h, 2t)? MHH_PRES_Y_TWISTED=1 / 0: wherever the plan has it / never
static bool pres_y_twisted(const mhh_pres_plan* P)
{
    if (!P->tw_ok || P->order != 2) return false;
    const char* e = getenv("MHH_PRES_Y_TWISTED");
    if (e) return !strcmp(e, "1");
    // measured on MI355X (profiles/r3_pres_forms.md): with fewer columns than the chip has CUs the stage is a dependent chain per block
    // and the half-length chains win (itot = 256: 0.245 -> 0.160 ms at 256^3, 0.511 -> 0.390 at 256 x 256 x 512); at itot = 512 there
    // is a block per CU already and the second launch costs more than it saves (0.93 -> 1.07 ms at 512^3)
    return P->itot <= 256;
}
static int lds_levels_per_block(const mhh_pres_plan* P)
{
    // enough blocks to fill the chip several times over (8 rows x kc levels each); every block re-does one level below its own
    const char* e = getenv("MHH_PRES_LDS_KC");
    int kc = e ? atoi(e) : (int)(((long long)P->ktot * (P->jtot/8)) / 2048);
    if (!e) kc = kc < 4 ? 4 : (kc > 32 ? 32 : kc);
    return kc < 1 ? 1 : (kc > P->ktot ? P->ktot : kc);
}
// Pres_4::output's correction of wt (src/pres_4.cxx:565-569; k = kstart is skipped) from p with its ghost levels in place: the
// part of the LDS form's last stage that needs the level above (pres_lds4.h)
template<class TF>
struct Pres4WtOp
{
    GridDev<TF> g; const TF* __restrict__ p; TF* __restrict__ wt;
    __device__ void operator()(int, int, int k, int c) const
    {
        const int kk = g.ijcells;
        wt[c] -= cg4(p[c-2*kk], p[c-kk], p[c], p[c+kk]) * g.dzhi4[k];
    }
};
// the same, one thread per column and chunk of levels with the four levels of p in a sliding window: one read of p per cell (+ three
// per chunk) where the cell form reads four (0.185 -> see profiles/r3_pres_forms.md at 512 x 256 x 256)
template<class TF>
__global__ void __launch_bounds__(256) pres4_wt_march_kernel(const GridDev<TF> g, const TF* __restrict__ p, TF* __restrict__ wt, int kc)
{
    const int i = g.istart + blockIdx.x*64 + threadIdx.x, j = g.jstart + blockIdx.y*4 + threadIdx.y;
    const int k0 = g.kstart + 1 + blockIdx.z*kc, k1 = (k0 + kc < g.kend) ? k0 + kc : g.kend;
    if (i >= g.iend || j >= g.jend || k0 >= k1) return;
    const int kk = g.ijcells;
    int c = i + j*g.icells + k0*kk;
    TF pm2 = p[c-2*kk], pm1 = p[c-kk], pc = p[c];
    for (int k=k0; k<k1; ++k, c+=kk)
    {
        const TF pn = p[c+kk];
        wt[c] -= cg4(pm2, pm1, pc, pn) * uniform_load(g.dzhi4, k);
        pm2 = pm1; pm1 = pc; pc = pn;
    }
}
static int pres4_lds_stage(mhh_pres_plan* P, const mhh_grid* g, const mhh_fields* f, double dt, int stage, dim3 xgrid, int kc, hipStream_t st)
{
    const int nx = ilog2(P->itot/2), ny = ilog2(P->jtot);
    if (stage == 1)
    {
        MHH_REQUIRE(f && f->u && f->v && f->w && f->ut && f->vt && f->wt, "null field");
        MHH_REQUIRE(dt > 0., "dt");
        if (int e = pres_input_halos(g, 4, f, st)) return e;
#define M(TF, N) else if (nx == N) hipLaunchKernelGGL((lds_fft::pres4_in_fftx_kernel<TF, (2 << N), N>), xgrid, dim3(P->itot), lds_bytes_x(P, 8), st, a);
#define CALL(TF) [&]{ lds_fft::Pres4LdsIn<TF> a{make_grid<TF>(g), cp<TF>(f->u), cp<TF>(f->v), cp<TF>(f->w), cp<TF>(f->ut), cp<TF>(f->vt), cp<TF>(f->wt), \
                          TF(1.)/TF(dt), static_cast<C2<TF>*>(P->spec), static_cast<const C2<TF>*>(P->tx), nx, kc}; \
                      if (P->itot <= LDS_XS) hipLaunchKernelGGL((lds_fft::pres4_in_fftx_kernel<TF, LDS_XS, 0>), xgrid, dim3(P->itot), lds_bytes_x(P, 8), st, a); \
                      MHH_FOR_NX_T(M, TF) return MHH_OK; }()
        if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
#undef M
    }
    else if (stage == 2)
    {
#define M(TF, N) else if (ny == N) { if constexpr (lds_has_ny<TF, N>()) hipLaunchKernelGGL((lds_fft::pres4_ysolve_kernel<TF, (1 << N), N>), yg, yb, yl, st, ya); }
#define CALL(TF) [&]{ const dim3 yg(P->itot/2), yb(P->jtot); const size_t yl = lds_bytes_y(P); \
                      const lds_fft::Pres4LdsSolve<TF> ya{static_cast<C2<TF>*>(P->spec), cp<TF>(P->w3l), cp<TF>(P->m[6]), static_cast<const C2<TF>*>(P->ty), P->itot/2, P->jtot, ny, P->ktot}; \
                      if (P->jtot <= LDS_YS) hipLaunchKernelGGL((lds_fft::pres4_ysolve_kernel<TF, LDS_YS, 0>), yg, yb, yl, st, ya); \
                      MHH_FOR_NY_T(M, TF) return MHH_OK; }()
        if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
#undef M
    }
    else
    {
        MHH_REQUIRE(f && f->p && f->ut && f->vt && f->wt, "null field");
#define M(TF, N) else if (nx == N) hipLaunchKernelGGL((lds_fft::pres4_ifftx_out_kernel<TF, (2 << N), N>), xgrid, dim3(P->itot), lds_bytes_x(P, 11), st, a);
#define CALL(TF) [&]{ lds_fft::Pres4LdsOut<TF> a{make_grid<TF>(g), static_cast<const C2<TF>*>(P->spec), static_cast<const C2<TF>*>(P->tx), \
                          mp<TF>(f->p), mp<TF>(f->ut), mp<TF>(f->vt), nx, kc}; \
                      if (P->itot <= LDS_XS) hipLaunchKernelGGL((lds_fft::pres4_ifftx_out_kernel<TF, LDS_XS, 0>), xgrid, dim3(P->itot), lds_bytes_x(P, 11), st, a); \
                      MHH_FOR_NX_T(M, TF) return MHH_OK; }()
        if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
#undef M
        MHH_LAUNCH_CHECK();
        const char* wm = getenv("MHH_PRES4_WT");                             // "cell": the one-thread-per-cell form (A/B)
        if (wm && !strcmp(wm, "cell"))
        {
#define CALL(TF) [&]{ const GridDev<TF> gd = make_grid<TF>(g); Pres4WtOp<TF> op{gd, cp<TF>(f->p), mp<TF>(f->wt)}; return launch_interior(st, gd, g->kstart + 1, g->kend, op); }()
            if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
        }
        else if (g->kmax > 1)
        {
            const int wkc = 32;
            const dim3 wg((g->imax + 63)/64, (g->jmax + 3)/4, (g->kmax - 1 + wkc - 1)/wkc);
#define CALL(TF) [&]{ hipLaunchKernelGGL(pres4_wt_march_kernel<TF>, wg, dim3(64, 4), 0, st, make_grid<TF>(g), cp<TF>(f->p), mp<TF>(f->wt), wkc); return MHH_OK; }()
            if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
        }
    }
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
// the three stages, also callable one by one (tests): 1 = input + x transform, 2 = y transforms + Thomas, 3 = x transform + p + output
MHH_API int mhh_pres_lds_stage(mhh_pres_plan* P, const mhh_grid* g, const mhh_fields* f, double dt, int stage, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(P && P->lds_ok, "plan has no LDS-transform form (power-of-two itot and jtot)");
    MHH_REQUIRE(P->dtype == g->dtype && P->itot == g->itot && P->jtot == g->jtot && P->ktot == g->ktot && g->npy == 1, "plan/grid mismatch");
    MHH_REQUIRE(stage >= 1 && stage <= 3, "stage");
    hipStream_t st = as_stream(stream);
    const int kc = lds_levels_per_block(P);
    const dim3 xgrid((unsigned)(P->jtot/8) * (unsigned)((P->ktot + kc-1)/kc));        // strips x chunks, decoded by lds_strip_of_block
    const int nx = ilog2(P->itot/2), ny = ilog2(P->jtot);
    if (P->order == 4) return pres4_lds_stage(P, g, f, dt, stage, xgrid, kc, st);
    if (stage == 1)
    {
        MHH_REQUIRE(f && f->u && f->v && f->w && f->ut && f->vt && f->wt && f->rhoref && f->rhorefh, "null field");
        MHH_REQUIRE(dt > 0., "dt");
        if (int e = pres_input_halos(g, 2, f, stream)) return e;
#define M(TF, N) else if (nx == N) hipLaunchKernelGGL((lds_fft::pres_in_fftx_kernel<TF, LDS_RG, (2 << N), N>), xgrid, dim3(P->itot), lds_bytes_x(P, 8), st, a);
#define CALL(TF) [&]{ lds_fft::PresLdsIn<TF> a{make_grid<TF>(g), cp<TF>(f->u), cp<TF>(f->v), cp<TF>(f->w), cp<TF>(f->ut), cp<TF>(f->vt), cp<TF>(f->wt), \
                          cp<TF>(f->rhoref), cp<TF>(f->rhorefh), TF(1.)/TF(dt), static_cast<C2<TF>*>(P->spec), static_cast<const C2<TF>*>(P->tx), nx, kc, 0, P->ktot, P->jtot, {}}; \
                      if (P->itot <= LDS_XS) hipLaunchKernelGGL((lds_fft::pres_in_fftx_kernel<TF, LDS_RG, LDS_XS, 0>), xgrid, dim3(P->itot), lds_bytes_x(P, 8), st, a); \
                      MHH_FOR_NX_T(M, TF) return MHH_OK; }()
        if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
#undef M
    }
    else if (stage == 2 && pres_y_twisted(P))
    {
        // two blocks per column, two launches: eliminate towards the split | solve where the halves meet and substitute outward
#define M(TF, N) else if (ny == N) { if constexpr (lds_has_ny<TF, N>()) { hipLaunchKernelGGL((lds_fft::pres_ysolve_tw_kernel<TF, (1 << N), N, 1>), yg, yb, yl, st, ya); \
                                                                          hipLaunchKernelGGL((lds_fft::pres_ysolve_tw_kernel<TF, (1 << N), N, 2>), yg, yb, yl, st, ya); } }
#define CALL(TF) [&]{ const dim3 yg(P->itot), yb(P->jtot); const size_t yl = lds_bytes_y(P); \
                      const lds_fft::PresLdsSolveTw<TF> ya{lds_solve_args<TF>(P), cp<TF>(P->a3l), static_cast<C2<TF>*>(P->itw), P->ksplit}; \
                      if (P->jtot <= LDS_YS) { hipLaunchKernelGGL((lds_fft::pres_ysolve_tw_kernel<TF, LDS_YS, 0, 1>), yg, yb, yl, st, ya); \
                                               hipLaunchKernelGGL((lds_fft::pres_ysolve_tw_kernel<TF, LDS_YS, 0, 2>), yg, yb, yl, st, ya); } \
                      MHH_FOR_NY_T(M, TF) return MHH_OK; }()
        if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
#undef M
    }
    else if (stage == 2)
    {
#define M(TF, N) else if (ny == N) { if constexpr (lds_has_ny<TF, N>()) hipLaunchKernelGGL((lds_fft::pres_ysolve_kernel<TF, (1 << N), N>), yg, yb, yl, st, ya); }
#define CALL(TF) [&]{ const dim3 yg(P->itot/2), yb(P->jtot); const size_t yl = lds_bytes_y(P); const auto ya = lds_solve_args<TF>(P); \
                      if (P->jtot <= LDS_YS) hipLaunchKernelGGL((lds_fft::pres_ysolve_kernel<TF, LDS_YS, 0>), yg, yb, yl, st, ya); \
                      MHH_FOR_NY_T(M, TF) return MHH_OK; }()
        if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
#undef M
    }
    else
    {
        MHH_REQUIRE(f && f->p && f->ut && f->vt && f->wt, "null field");
#define M(TF, N) else if (nx == N) { if (P->rk_on) hipLaunchKernelGGL((lds_fft::pres_ifftx_out_kernel<TF, LDS_RG, (2 << N), N, false, true>), xgrid, dim3(P->itot), lds_bytes_x(P, 9), st, a); \
                                    else          hipLaunchKernelGGL((lds_fft::pres_ifftx_out_kernel<TF, LDS_RG, (2 << N), N>), xgrid, dim3(P->itot), lds_bytes_x(P, 9), st, a); }
#define CALL(TF) [&]{ lds_fft::PresLdsOut<TF> a{make_grid<TF>(g), static_cast<const C2<TF>*>(P->spec), static_cast<const C2<TF>*>(P->tx), \
                          mp<TF>(f->p), mp<TF>(f->ut), mp<TF>(f->vt), mp<TF>(f->wt), nx, kc, 0, P->ktot, P->jtot, {}, \
                          mp<TF>(P->rk_u), mp<TF>(P->rk_v), mp<TF>(P->rk_w), TF(P->rk_cA), TF(P->rk_cB), TF(P->rk_dt)}; \
                      if (P->itot <= LDS_XS) { if (P->rk_on) hipLaunchKernelGGL((lds_fft::pres_ifftx_out_kernel<TF, LDS_RG, LDS_XS, 0, false, true>), xgrid, dim3(P->itot), lds_bytes_x(P, 9), st, a); \
                                               else          hipLaunchKernelGGL((lds_fft::pres_ifftx_out_kernel<TF, LDS_RG, LDS_XS, 0>), xgrid, dim3(P->itot), lds_bytes_x(P, 9), st, a); } \
                      MHH_FOR_NX_T(M, TF) return MHH_OK; }()
        if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
#undef M
    }
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
// ---- the x stages for a slab rank (k_slab.hip): the same kernels on the rank's own rows, reading / writing the all-to-all
// buffers of the x <-> y transposes directly (pres_lds.h, LdsSlab). Internal to the library (pres_lds_slab.h).
namespace mhh
{
template<class TF>
static int lds_slab_kernels_ready(int itot, bool& usable)
{
    const int nx = ilog2(itot/2);
    const void* k[2] = {nullptr, nullptr};
    if (itot <= LDS_XS) { k[0] = reinterpret_cast<const void*>(&lds_fft::pres_in_fftx_kernel<TF, LDS_RG, LDS_XS, 0, true>); k[1] = reinterpret_cast<const void*>(&lds_fft::pres_ifftx_out_kernel<TF, LDS_RG, LDS_XS, 0, true>); }
#define M(N) if (nx == N) { k[0] = reinterpret_cast<const void*>(&lds_fft::pres_in_fftx_kernel<TF, LDS_RG, (2 << N), N, true>); k[1] = reinterpret_cast<const void*>(&lds_fft::pres_ifftx_out_kernel<TF, LDS_RG, (2 << N), N, true>); }
    MHH_FOR_NX(M)
#undef M
    usable = true;
    for (int n=0; n<2; ++n)
    {
        if (!k[n]) { usable = false; return MHH_OK; }
        bool u = false;
        if (int e = lds_kernel_ready(k[n], u)) return e;
        usable = usable && u;
    }
    return MHH_OK;
}
template<class TF>
static int lds_slab_y_ready(int jtot, bool& usable)
{
    const int ny = ilog2(jtot);
    const void* k[2] = {nullptr, nullptr};
    if (jtot <= LDS_YS) { k[0] = reinterpret_cast<const void*>(&lds_fft::slab_yfft_kernel<TF, LDS_YS, 0, true>); k[1] = reinterpret_cast<const void*>(&lds_fft::slab_yfft_kernel<TF, LDS_YS, 0, false>); }
#define M(N) if (ny == N) { if constexpr (lds_has_ny<TF, N>()) { k[0] = reinterpret_cast<const void*>(&lds_fft::slab_yfft_kernel<TF, (1 << N), N, true>); k[1] = reinterpret_cast<const void*>(&lds_fft::slab_yfft_kernel<TF, (1 << N), N, false>); } }
    MHH_FOR_NY(M)
#undef M
    usable = true;
    for (int n=0; n<2; ++n)
    {
        if (!k[n]) { usable = false; return MHH_OK; }
        bool u = false;
        if (int e = lds_kernel_ready(k[n], u)) return e;
        usable = usable && u;
    }
    return MHH_OK;
}
static size_t lds_slab_bytes_x(int itot, size_t esz, int rows) { return ((size_t)rows*(itot/2 + 2) + itot) * 2*esz + (rows == 9 ? (size_t)8*itot*esz : 0); }
// 1 if a rank of this grid can run the x stages in LDS: power-of-two itot with an instantiation, rows in whole strips of eight
int lds_slab_usable(const mhh_grid* g)
{
    if (!(is_pow2(g->itot) && g->itot >= 16 && g->itot <= 1024 && g->jmax % 8 == 0 && g->jmax >= 8)) return 0;
    if (lds_slab_bytes_x(g->itot, g->dtype == MHH_F64 ? 8 : 4, 9) > 160*1024 || g->igc > g->itot) return 0;
    if ((long long)g->icells*g->jcells*g->kcells >= (1ll << 31)) return 0;
    if (!(is_pow2(g->jtot) && g->jtot >= 8 && g->jtot <= 1024 && g->jtot % g->jmax == 0)) return 0;
    if (((size_t)8*g->jtot + g->jtot) * 2*(g->dtype == MHH_F64 ? 8 : 4) > 160*1024) return 0;
    bool usable = false, usable_y = false;
    int e = (g->dtype == MHH_F64) ? lds_slab_kernels_ready<double>(g->itot, usable) : lds_slab_kernels_ready<float>(g->itot, usable);
    if (e == MHH_OK) e = (g->dtype == MHH_F64) ? lds_slab_y_ready<double>(g->jtot, usable_y) : lds_slab_y_ready<float>(g->jtot, usable_y);
    return (e == MHH_OK && usable && usable_y) ? 1 : 0;
}
// exp(-2 pi i m / itot), m < itot, on the device (the caller frees it)
int lds_slab_twiddles(const mhh_grid* g, void** tx)
{
    const double pi = std::acos(-1.);
    const int n = g->itot;
    auto fill = [&](auto* t) { for (int m=0; m<n; ++m) { t[2*m] = std::cos(2.*pi*m/n); t[2*m+1] = -std::sin(2.*pi*m/n); }
                               t[0] = 1; t[1] = 0; t[n] = -1; t[n+1] = 0; if (n >= 4) { t[n/2] = 0; t[n/2+1] = -1; t[3*n/2] = 0; t[3*n/2+1] = 1; } };
    if (g->dtype == MHH_F64) { std::vector<double> t(2*(size_t)n); fill(t.data()); return upload(tx, t); }
    std::vector<float> t(2*(size_t)n); fill(t.data()); return upload(tx, t);
}
// exp(-2 pi i m / jtot), m < jtot
int lds_slab_twiddles_y(const mhh_grid* g, void** ty)
{
    const double pi = std::acos(-1.);
    const int n = g->jtot;
    auto fill = [&](auto* t) { for (int m=0; m<n; ++m) { t[2*m] = std::cos(2.*pi*m/n); t[2*m+1] = -std::sin(2.*pi*m/n); }
                               t[0] = 1; t[1] = 0; t[n] = -1; t[n+1] = 0; if (n >= 4) { t[n/2] = 0; t[n/2+1] = -1; t[3*n/2] = 0; t[3*n/2+1] = 1; } };
    if (g->dtype == MHH_F64) { std::vector<double> t(2*(size_t)n); fill(t.data()); return upload(ty, t); }
    std::vector<float> t(2*(size_t)n); fill(t.data()); return upload(ty, t);
}
// the transforms along y of the levels [kbeg, kend): forward from the receive buffer into specy[k][kxl][ky], or back from specy into the send buffer
int lds_slab_yfft(const mhh_grid* g, bool fwd, void* xbuf, void* specy, const void* ty, int nxb, int npy, int ks, int kbeg, int kend, hipStream_t st)
{
    const int ny = ilog2(g->jtot);
    const dim3 grid((unsigned)nxb, (unsigned)((kend - kbeg + 7)/8)), block((unsigned)g->jtot);
    const size_t lds = ((size_t)8*g->jtot + g->jtot) * 2*(g->dtype == MHH_F64 ? 8 : 4);
#define M(TF, N) else if (ny == N) { if constexpr (lds_has_ny<TF, N>()) { \
                     if (fwd) hipLaunchKernelGGL((lds_fft::slab_yfft_kernel<TF, (1 << N), N, true>), grid, block, lds, st, a); \
                     else     hipLaunchKernelGGL((lds_fft::slab_yfft_kernel<TF, (1 << N), N, false>), grid, block, lds, st, a); } }
#define CALL(TF) [&]{ lds_fft::SlabYfft<TF> a{static_cast<C2<TF>*>(xbuf), static_cast<C2<TF>*>(specy), static_cast<const C2<TF>*>(ty), g->jtot, g->jmax, ny, nxb, kbeg, kend, {nxb, npy, ks}}; \
                      if (g->jtot <= LDS_YS) { if (fwd) hipLaunchKernelGGL((lds_fft::slab_yfft_kernel<TF, LDS_YS, 0, true>), grid, block, lds, st, a); \
                                               else     hipLaunchKernelGGL((lds_fft::slab_yfft_kernel<TF, LDS_YS, 0, false>), grid, block, lds, st, a); } \
                      MHH_FOR_NY_T(M, TF) return MHH_OK; }()
    if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
#undef M
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
static int lds_slab_kc(const mhh_grid* g, int nlev)
{
    const char* e = getenv("MHH_PRES_LDS_KC");
    int kc = e ? atoi(e) : (int)(((long long)nlev * (g->jmax/8)) / 2048);
    if (!e) kc = kc < 4 ? 4 : (kc > 32 ? 32 : kc);
    return kc < 1 ? 1 : (kc > nlev ? nlev : kc);
}
// Pres_2::input + the transform along x of the levels [kbeg, kend) into the send buffer of the x -> y transpose
int lds_slab_stage_in(const mhh_grid* g, const mhh_fields* f, double dt, void* xbuf, const void* tx, int nxb, int npy, int ks, int kbeg, int kend, hipStream_t st)
{
    if (int e = pres_input_halos(g, 2, f, st)) return e;
    const int nx = ilog2(g->itot/2), kc = lds_slab_kc(g, kend - kbeg);
    const dim3 xgrid((unsigned)(g->jmax/8) * (unsigned)((kend - kbeg + kc-1)/kc));
    const size_t lds = lds_slab_bytes_x(g->itot, g->dtype == MHH_F64 ? 8 : 4, 8);
#define M(TF, N) else if (nx == N) hipLaunchKernelGGL((lds_fft::pres_in_fftx_kernel<TF, LDS_RG, (2 << N), N, true>), xgrid, dim3(g->itot), lds, st, a);
#define CALL(TF) [&]{ lds_fft::PresLdsIn<TF> a{make_grid<TF>(g), cp<TF>(f->u), cp<TF>(f->v), cp<TF>(f->w), cp<TF>(f->ut), cp<TF>(f->vt), cp<TF>(f->wt), \
                          cp<TF>(f->rhoref), cp<TF>(f->rhorefh), TF(1.)/TF(dt), static_cast<C2<TF>*>(xbuf), static_cast<const C2<TF>*>(tx), nx, kc, kbeg, kend, g->jmax, {nxb, npy, ks}}; \
                      if (g->itot <= LDS_XS) hipLaunchKernelGGL((lds_fft::pres_in_fftx_kernel<TF, LDS_RG, LDS_XS, 0, true>), xgrid, dim3(g->itot), lds, st, a); \
                      MHH_FOR_NX_T(M, TF) return MHH_OK; }()
    if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
#undef M
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
// the transform back along x of the levels [kbeg, kend) from the receive buffer of the y -> x transpose + p + Pres_2::output
int lds_slab_stage_out(const mhh_grid* g, const mhh_fields* f, const void* xbuf, const void* tx, int nxb, int npy, int ks, int kbeg, int kend, hipStream_t st)
{
    const int nx = ilog2(g->itot/2), kc = lds_slab_kc(g, kend - kbeg);
    const dim3 xgrid((unsigned)(g->jmax/8) * (unsigned)((kend - kbeg + kc-1)/kc));
    const size_t lds = lds_slab_bytes_x(g->itot, g->dtype == MHH_F64 ? 8 : 4, 9);
#define M(TF, N) else if (nx == N) hipLaunchKernelGGL((lds_fft::pres_ifftx_out_kernel<TF, LDS_RG, (2 << N), N, true>), xgrid, dim3(g->itot), lds, st, a);
#define CALL(TF) [&]{ lds_fft::PresLdsOut<TF> a{make_grid<TF>(g), static_cast<const C2<TF>*>(xbuf), static_cast<const C2<TF>*>(tx), \
                          mp<TF>(f->p), mp<TF>(f->ut), mp<TF>(f->vt), mp<TF>(f->wt), nx, kc, kbeg, kend, g->jmax, {nxb, npy, ks}, nullptr, nullptr, nullptr, TF(0), TF(0), TF(0)}; \
                      if (g->itot <= LDS_XS) hipLaunchKernelGGL((lds_fft::pres_ifftx_out_kernel<TF, LDS_RG, LDS_XS, 0, true>), xgrid, dim3(g->itot), lds, st, a); \
                      MHH_FOR_NX_T(M, TF) return MHH_OK; }()
    if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
#undef M
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
} // namespace mhh

// 1 if the plan can run the LDS-transform form (mhh_pres_exec takes it by itself on large grids, see there)
MHH_API int mhh_pres_plan_has_lds_form(const mhh_pres_plan* P) { return (P && P->lds_ok) ? 1 : 0; }
// the spectral array between the stages (tests): S[k][kx][j], complex
MHH_API void* mhh_pres_plan_spectral(mhh_pres_plan* P) { return P ? P->spec : nullptr; }

// the form mhh_pres_exec takes for this plan: 0 = staged (rocFFT), 1 = transforms in LDS
MHH_API int mhh_pres_exec_form(const mhh_pres_plan* P)
{
    if (!P || !P->lds_ok) return 0;
    const char* le = getenv("MHH_PRES_LDS");
    // measured on MI355X with every row length in its own instantiation (profiles/r3_pres_forms.md; ms staged / LDS form):
    // 128^3 0.137 / 0.184, 256^3 0.742 / 0.688, 256x256x512 1.55 / 1.43, 512x256x256 1.38 / 1.16, 512x512x128 1.44 / 1.12,
    // 512^3 5.80 / 4.05, 1024x512x256 6.00 / 4.60; fp32 256^3 0.472 / 0.464, 512^3 3.68 / 2.42, 1024x1024x256 7.81 / 4.61:
    // the LDS form from 2^24 cells on (below that the arrays sit in the Infinity Cache and the staged passes are cheap)
    const bool lds_large = (long long)P->itot*P->jtot*P->ktot >= (1ll << 24);
    return (le ? !strcmp(le, "1") : lds_large) ? 1 : 0;
}
MHH_API int mhh_pres_exec(mhh_pres_plan* P, const mhh_grid* g, const mhh_fields* f, double dt, void* stream)
{
    MHH_REQUIRE(P != nullptr, "plan");
    // Opt-in (MHH_PRES_FUSED=1). Measured on MI355X: identical bits, but the transforms run the heavy producer through an
    // indirect call per element and lose more than the two saved array passes give back (512^3: 15.7 ms vs 14.7 ms per step;
    // 512x256x256 pres_4: 4.05 vs 3.42 ms), so the staged form stays the default.
    // The LDS form pays where the arrays are far larger than the caches and there is a block per CU for the y stage (measured on
    // MI355X: 512^3 fp64 5.8 -> 5.0 ms, 1024 x 1024 x 256 fp32 7.9 -> 5.0 ms; 256^3 0.78 -> 0.92 ms, so not there).
    // MHH_PRES_LDS=0 / 1: never / wherever the plan has the form.
    if (mhh_pres_exec_form(P) == 1)
    {
        for (int stage=1; stage<=3; ++stage) if (int e = mhh_pres_lds_stage(P, g, f, dt, stage, stream)) return e;
        return MHH_OK;
    }
    const char* env = getenv("MHH_PRES_FUSED");
    if (!P->cb_ready || !(env && !strcmp(env, "1")))
    {
        if (int e = mhh_pres_input(P, g, f, dt, nullptr, stream)) return e;
        const char* uo = getenv("MHH_PRES_UNPACK_OUT");                       // "0": unpack and output as two kernels (A/B)
        if (P->order == 2 && !(uo && !strcmp(uo, "0")))
        {
            if (int e = check_grid(g)) return e;
            MHH_REQUIRE(f && f->p && f->ut && f->vt && f->wt, "null field");
            MHH_REQUIRE(P->dtype == g->dtype && P->itot == g->itot && P->jtot == g->jtot && P->ktot == g->ktot, "plan/grid mismatch");
            hipStream_t st = as_stream(stream);
            if (int e = pres_spectral(P, g, P->packed, st)) return e;
            dim3 ug((g->icells*g->jcells + 255)/256, 1, g->kmax + 1);
            const bool pow2 = is_pow2(g->itot) && is_pow2(g->jtot);
#define RKARGS(TF) (P->rk_on ? mp<TF>(P->rk_u) : nullptr), mp<TF>(P->rk_v), mp<TF>(P->rk_w), TF(P->rk_cA), TF(P->rk_cB), TF(P->rk_dt)
#define CALL(TF) [&]{ const GridDev<TF> gd = make_grid<TF>(g); \
                if (pow2) hipLaunchKernelGGL((unpack_out2_kernel<true, TF>), ug, dim3(256), 0, st, mp<TF>(f->p), cp<TF>(P->packed), mp<TF>(f->ut), mp<TF>(f->vt), mp<TF>(f->wt), gd.dzhi, \
                                   gd.dxi_t, gd.dyi_t, g->itot, g->jtot, g->kmax, g->igc, g->jgc, g->kgc, g->icells, g->jcells, RKARGS(TF)); \
                else hipLaunchKernelGGL((unpack_out2_kernel<false, TF>), ug, dim3(256), 0, st, mp<TF>(f->p), cp<TF>(P->packed), mp<TF>(f->ut), mp<TF>(f->vt), mp<TF>(f->wt), gd.dzhi, \
                                   gd.dxi_t, gd.dyi_t, g->itot, g->jtot, g->kmax, g->igc, g->jgc, g->kgc, g->icells, g->jcells, RKARGS(TF)); return MHH_OK; }()
            if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
            MHH_LAUNCH_CHECK();
            return MHH_OK;
        }
        if (P->order == 4 && !(uo && !strcmp(uo, "0")))
        {
            if (int e = check_grid(g)) return e;
            MHH_REQUIRE(f && f->p && f->ut && f->vt && f->wt, "null field");
            MHH_REQUIRE(P->dtype == g->dtype && P->itot == g->itot && P->jtot == g->jtot && P->ktot == g->ktot, "plan/grid mismatch");
            hipStream_t st = as_stream(stream);
            if (int e = pres_spectral(P, g, P->packed, st)) return e;
            dim3 ug((g->icells*g->jcells + 255)/256, 1, g->kmax + 4);
            const bool pow2 = is_pow2(g->itot) && is_pow2(g->jtot);
#define CALL(TF) [&]{ const GridDev<TF> gd = make_grid<TF>(g); \
                if (pow2) hipLaunchKernelGGL((unpack_out4_kernel<true, TF>), ug, dim3(256), 0, st, mp<TF>(f->p), cp<TF>(P->packed), mp<TF>(f->ut), mp<TF>(f->vt), mp<TF>(f->wt), gd.dzhi4, \
                                   gd.dxi_d, gd.dyi_d, (int)gd.dim3, g->itot, g->jtot, g->kmax, g->igc, g->jgc, g->kgc, g->icells, g->jcells); \
                else hipLaunchKernelGGL((unpack_out4_kernel<false, TF>), ug, dim3(256), 0, st, mp<TF>(f->p), cp<TF>(P->packed), mp<TF>(f->ut), mp<TF>(f->vt), mp<TF>(f->wt), gd.dzhi4, \
                                   gd.dxi_d, gd.dyi_d, (int)gd.dim3, g->itot, g->jtot, g->kmax, g->igc, g->jgc, g->kgc, g->icells, g->jcells); return MHH_OK; }()
            if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
            MHH_LAUNCH_CHECK();
            return MHH_OK;
        }
        if (int e = mhh_pres_solve(P, g, f, nullptr, stream)) return e;
        return mhh_pres_output(P, g, f, stream);
    }
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(f && f->p && f->u && f->v && f->w && f->ut && f->vt && f->wt, "null field");
    MHH_REQUIRE(P->dtype == g->dtype && P->itot == g->itot && P->jtot == g->jtot && P->ktot == g->ktot, "plan/grid mismatch");
    MHH_REQUIRE(dt > 0., "dt");
    MHH_REQUIRE(P->order == 4 || (f->rhoref && f->rhorefh), "rhoref");
    MHH_REQUIRE((unsigned long long)g->itot*g->jtot*g->ktot < (1ull << 31), "fused form indexes the packed cells with 32 bits");
    if (int e = pres_input_halos(g, P->order, f, stream)) return e;
    if (int e = (g->dtype == MHH_F64) ? pres_exec_fused<double>(P, g, f, dt, as_stream(stream)) : pres_exec_fused<float>(P, g, f, dt, as_stream(stream))) return e;
    return mhh_pres_output(P, g, f, stream);
}

// Pres::exec followed by the Runge-Kutta sub-step of u, v, w (timeloop.exec() in Model::exec, src/model.cxx:411,484;
// src/timeloop.cxx:250-334): where the corrected tendencies are stored by a pres_2 kernel of this library (the LDS form's last stage,
// the staged form's unpack + output kernel) the sub-step is applied there, in registers; otherwise -- pres_4, the two-kernel and
// callback forms, the last sub-step of a step (the tendency reset covers the ghost cells) -- Pres::exec is followed by three
// mhh_rk_substep calls. Either way the bits of mhh_pres_exec + mhh_rk_substep x 3. Scalars keep their own mhh_rk_substep call.
MHH_API int mhh_pres_exec_rk(mhh_pres_plan* P, const mhh_grid* g, const mhh_fields* f, double dt, int rkorder, int substep, double rkdt, void* stream)
{
    MHH_REQUIRE(P != nullptr && f != nullptr && f->u && f->v && f->w && f->ut && f->vt && f->wt, "plan, fields");
    double cA = 0, cB = 0; bool reset = false;
    MHH_REQUIRE(rk_coefficients(rkorder, substep, cA, cB, reset), "rkorder 3 or 4, substep in range");
    const char* uo = getenv("MHH_PRES_UNPACK_OUT"); const char* fe = getenv("MHH_PRES_FUSED"); const char* rke = getenv("MHH_PRES_RK_FUSED");
    const bool one_kernel = mhh_pres_exec_form(P) == 1 || (P->order == 2 && !(uo && !strcmp(uo, "0")) && !(P->cb_ready && fe && !strcmp(fe, "1")));
    const bool fuse = !reset && P->order == 2 && one_kernel && !(rke && !strcmp(rke, "0"));
    if (fuse)
    {
        P->rk_on = true; P->rk_cA = cA; P->rk_cB = cB; P->rk_dt = rkdt; P->rk_u = f->u; P->rk_v = f->v; P->rk_w = f->w;
        const int e = mhh_pres_exec(P, g, f, dt, stream);
        P->rk_on = false;
        return e;
    }
    if (int e = mhh_pres_exec(P, g, f, dt, stream)) return e;
    if (int e = mhh_rk_substep(g, rkorder, substep, rkdt, f->u, f->ut, stream)) return e;
    if (int e = mhh_rk_substep(g, rkorder, substep, rkdt, f->v, f->vt, stream)) return e;
    return mhh_rk_substep(g, rkorder, substep, rkdt, f->w, f->wt, stream);
}
