// k_slab.hip -- the pieces of the hot path that differ when the grid is slab-decomposed in y over the GPUs of
// a node (npx = 1, npy = N): north-south halo pack/unpack around the neighbour exchange, and
// the pressure solver split at the x<->y transpose. The exchanges themselves (ring send/recv, all-to-all)
// are issued by the host through torch.distributed (RCCL over xGMI); nothing here communicates.
//
// Reference semantics: Boundary_cyclic::exec's MPI path (src/boundary_cyclic.cxx:116-176) and the FFT/solve
// sequence of src/fft.cxx:451-583 with npx = 1, where only Transpose::exec_xy / exec_yx move data
// (src/transpose.cxx:170-219) and Pres_2::solve swaps the mode indices (src/pres_2.cxx:297-299).
#include <vector>
#include <rocfft/rocfft.h>
#include "fft_lifetime.h"
#include "k_common.h"
#include "pres_lds_slab.h"

using namespace mhh;

#define MHH_FFT_TRY(expr) do { rocfft_status s_ = (expr); if (s_ != rocfft_status_success) { \
    mhh::set_error("FFT error: %s returned %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__); return MHH_EFFT; } } while (0)

// =======================================================================================================
// North-south halo: buffers are [field][k][jg][i] with all icells (x ghosts included, so corners are right
// once the east-west fill has run first, like the reference's ordering).
// =======================================================================================================
constexpr int MAXF = 8;
template<class TF> struct FieldList { TF* f[MAXF]; };

// rs rows travel south (my southernmost interior rows -> the first rs north ghost rows of the south neighbour), rn rows
// travel north (my northernmost interior rows -> the rn south ghost rows next to the north neighbour's interior).
// The full exchange is rs = rn = jgc; a consumer that reads one row beyond the slab in one direction only (pres input:
// vt[j+1]; pres output: p[j-1]) moves just that row.
template<class TF, bool PACK>
__global__ void __launch_bounds__(256) halo_ns_kernel(FieldList<TF> fl, TF* __restrict__ south, TF* __restrict__ north,
                                                      int icells, int ijcells, int kcells, int jgc, int jstart, int jend, int rs, int rn)
{
    const int i = blockIdx.x*256 + threadIdx.x;
    const int k = blockIdx.y, n = blockIdx.z;
    if (i >= icells) return;
    TF* __restrict__ a = fl.f[n] + (size_t)k*ijcells;
    const size_t bs = ((size_t)n*kcells + k) * rs * icells + i, bn = ((size_t)n*kcells + k) * rn * icells + i;
    if (PACK)
    {
        for (int j=0; j<rs; ++j) south[bs + (size_t)j*icells] = a[i + (jstart + j)*icells];
        for (int j=0; j<rn; ++j) north[bn + (size_t)j*icells] = a[i + (jend - rn + j)*icells];
    }
    else
    {
        // `south` holds what the south neighbour sent north (its rn rows), `north` what the north neighbour sent south (rs rows)
        for (int j=0; j<rn; ++j) a[i + (jgc - rn + j)*icells] = south[bn + (size_t)j*icells];
        for (int j=0; j<rs; ++j) a[i + (jend + j)*icells] = north[bs + (size_t)j*icells];
    }
}
template<class TF, bool PACK>
static int halo_launch(const mhh_grid* g, void* const* fields, int nf, void* south, void* north, int rs, int rn, hipStream_t st)
{
    FieldList<TF> fl;
    for (int n=0; n<MAXF; ++n) fl.f[n] = mp<TF>(fields[n < nf ? n : 0]);
    hipLaunchKernelGGL((halo_ns_kernel<TF, PACK>), dim3((g->icells + 255)/256, g->kcells, nf), dim3(256), 0, st, fl, mp<TF>(south), mp<TF>(north),
                       g->icells, g->ijcells, g->kcells, g->jgc, g->jstart, g->jend, rs, rn);
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
static int halo_check(const mhh_grid* g, void* const* fields, int nf, const void* a, const void* b)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(nf >= 1 && nf <= MAXF && fields && a && b, "1..8 fields, non-null buffers");
    for (int n=0; n<nf; ++n) MHH_REQUIRE(fields[n] != nullptr, "null field");
    MHH_REQUIRE(g->jmax >= g->jgc, "jmax >= jgc (src/grid.cxx:420)");
    return MHH_OK;
}
MHH_API int mhh_halo_pack_rows(const mhh_grid* g, void* const* fields, int nf, int rows_south, int rows_north, void* send_south, void* send_north, void* stream)
{
    if (int e = halo_check(g, fields, nf, send_south, send_north)) return e;
    MHH_REQUIRE(rows_south >= 0 && rows_south <= g->jgc && rows_north >= 0 && rows_north <= g->jgc && rows_south + rows_north > 0, "0 <= rows <= jgc");
    if (g->dtype == MHH_F64) return halo_launch<double, true>(g, fields, nf, send_south, send_north, rows_south, rows_north, as_stream(stream));
    return halo_launch<float, true>(g, fields, nf, send_south, send_north, rows_south, rows_north, as_stream(stream));
}
MHH_API int mhh_halo_unpack_rows(const mhh_grid* g, void* const* fields, int nf, int rows_south, int rows_north, const void* recv_south, const void* recv_north, void* stream)
{
    if (int e = halo_check(g, fields, nf, recv_south, recv_north)) return e;
    MHH_REQUIRE(rows_south >= 0 && rows_south <= g->jgc && rows_north >= 0 && rows_north <= g->jgc && rows_south + rows_north > 0, "0 <= rows <= jgc");
    if (g->dtype == MHH_F64) return halo_launch<double, false>(g, fields, nf, const_cast<void*>(recv_south), const_cast<void*>(recv_north), rows_south, rows_north, as_stream(stream));
    return halo_launch<float, false>(g, fields, nf, const_cast<void*>(recv_south), const_cast<void*>(recv_north), rows_south, rows_north, as_stream(stream));
}
MHH_API int mhh_halo_pack_ns(const mhh_grid* g, void* const* fields, int nf, void* send_south, void* send_north, void* stream)
{ return mhh_halo_pack_rows(g, fields, nf, g ? g->jgc : 0, g ? g->jgc : 0, send_south, send_north, stream); }
MHH_API int mhh_halo_unpack_ns(const mhh_grid* g, void* const* fields, int nf, const void* recv_south, const void* recv_north, void* stream)
{ return mhh_halo_unpack_rows(g, fields, nf, g ? g->jgc : 0, g ? g->jgc : 0, recv_south, recv_north, stream); }
MHH_API unsigned long long mhh_halo_buffer_elems(const mhh_grid* g, int nf)
{
    return (unsigned long long)nf * g->kcells * g->jgc * g->icells;
}

// =======================================================================================================
// Slab pressure solver (pres_2). Layouts, all complex interleaved, nxh = itot/2+1, nxb = ceil(nxh/npy):
//   specx [k][jl][kx]            after the local x transform (jl in this rank's jmax rows)
//   xbuf  [q][k][jl][kxl]        all-to-all buffer, q = destination / source rank, kx = q*nxb + kxl (zero padded)
//   specy [k][kxl][j]            after the exchange, j over the full jtot, unit stride for the y transform
// =======================================================================================================
template<class TF> struct alignas(2*sizeof(TF)) C2 { TF x, y; };   // naturally aligned: one 16-byte access per fp64 number

struct mhh_pres_slab_plan
{
    int dtype = 0, itot = 0, jtot = 0, ktot = 0, jmax = 0, npy = 1, rank = 0, nxh = 0, nxb = 0;
    size_t esz = 8;
    void* bmati = nullptr; void* bmatj = nullptr; void* a = nullptr; void* c = nullptr; void* dz = nullptr; void* rhoref = nullptr;
    void* packed = nullptr; void* specx = nullptr; void* specy = nullptr; void* work = nullptr;
    rocfft_plan fx = nullptr, bx = nullptr, fy = nullptr, by = nullptr;
    // the same transforms over ONE k-slice of ktot / nchunks levels (mhh_pres_slab_set_chunks): slice c of the all-to-all can
    // travel while slice c+1 is transformed
    int nchunks = 1;
    rocfft_plan cfx = nullptr, cbx = nullptr, cfy = nullptr, cby = nullptr;
    rocfft_execution_info info = nullptr;
    void* wb = nullptr; size_t wbs = 0, wb_cap = 0;
    // the x stages with the transforms in LDS (pres_lds.h, pres_lds_slab.h): twiddles exp(-2 pi i m / itot); null = not available
    void* tx_lds = nullptr; void* ty_lds = nullptr;
};

template<class TF>
static int up(void** dst, const std::vector<TF>& v)
{
    MHH_HIP_TRY(hipMalloc(dst, v.size()*sizeof(TF)));
    MHH_HIP_TRY(hipMemcpy(*dst, v.data(), v.size()*sizeof(TF), hipMemcpyHostToDevice));
    return MHH_OK;
}
template<class TF>
static int slab_tables(mhh_pres_slab_plan* P, const mhh_grid* g, const void* hdz, const void* hdzhi, const void* hrho, const void* hrhoh)
{
    // Pres_2::set_values, src/pres_2.cxx:125-153
    const int itot = g->itot, jtot = g->jtot, kmax = g->kmax, kgc = g->kgc;
    const TF dx = TF(g->dx), dy = TF(g->dy);
    const TF dxidxi = 1./(dx*dx), dyidyi = 1./(dy*dy);
    const TF pi = std::acos(-1.);
    std::vector<TF> bi(itot), bj(jtot), a(kmax), c(kmax), dzk(kmax), rk(kmax);
    for (int j=0; j<jtot/2+1; ++j) bj[j] = 2. * (std::cos(2.*pi*(TF)j/(TF)jtot)-1.) * dyidyi;
    for (int j=jtot/2+1; j<jtot; ++j) bj[j] = bj[jtot-j];
    for (int i=0; i<itot/2+1; ++i) bi[i] = 2. * (std::cos(2.*pi*(TF)i/(TF)itot)-1.) * dxidxi;
    for (int i=itot/2+1; i<itot; ++i) bi[i] = bi[itot-i];
    const TF* dz = cp<TF>(hdz); const TF* dzhi = cp<TF>(hdzhi); const TF* rhoh = cp<TF>(hrhoh); const TF* rho = cp<TF>(hrho);
    for (int k=0; k<kmax; ++k)
    {
        a[k] = dz[k+kgc] * rhoh[k+kgc  ]*dzhi[k+kgc  ];
        c[k] = dz[k+kgc] * rhoh[k+kgc+1]*dzhi[k+kgc+1];
        dzk[k] = dz[k+kgc]; rk[k] = rho[k+kgc];
    }
    if (int e = up(&P->bmati, bi)) return e;
    if (int e = up(&P->bmatj, bj)) return e;
    if (int e = up(&P->a, a)) return e;
    if (int e = up(&P->c, c)) return e;
    if (int e = up(&P->dz, dzk)) return e;
    return up(&P->rhoref, rk);
}

static int plan1d(rocfft_plan* plan, rocfft_transform_type type, rocfft_result_placement place, int dtype, size_t n, size_t batch,
                  rocfft_array_type in_t, rocfft_array_type out_t, size_t in_dist, size_t out_dist, size_t* wbs)
{
    const size_t len[1] = {n}, one[1] = {1}, off[1] = {0};
    rocfft_plan_description d = nullptr;
    MHH_FFT_TRY(rocfft_plan_description_create(&d));
    MHH_FFT_TRY(rocfft_plan_description_set_data_layout(d, in_t, out_t, off, off, 1, one, in_dist, 1, one, out_dist));
    MHH_FFT_TRY(rocfft_plan_create(plan, place, type, dtype == MHH_F64 ? rocfft_precision_double : rocfft_precision_single, 1, len, batch, d));
    MHH_FFT_TRY(rocfft_plan_description_destroy(d));
    size_t w = 0;
    MHH_FFT_TRY(rocfft_plan_get_work_buffer_size(*plan, &w));
    if (w > *wbs) *wbs = w;
    return MHH_OK;
}

MHH_API void mhh_pres_slab_plan_destroy(mhh_pres_slab_plan* P)
{
    if (P && P->tx_lds) { (void)hipFree(P->tx_lds); P->tx_lds = nullptr; }
    if (P && P->ty_lds) { (void)hipFree(P->ty_lds); P->ty_lds = nullptr; }
    if (!P) return;
    for (rocfft_plan p : {P->fx, P->bx, P->fy, P->by, P->cfx, P->cbx, P->cfy, P->cby}) if (p) rocfft_plan_destroy(p);
    if (P->info) rocfft_execution_info_destroy(P->info);
    for (void* b : {P->bmati, P->bmatj, P->a, P->c, P->dz, P->rhoref, P->packed, P->specx, P->specy, P->work, P->wb}) if (b) (void)hipFree(b);
    delete P;
}

static int slab_factor(mhh_pres_slab_plan* P);     // tdma_slab_factor_kernel launch, defined with the kernel
MHH_API int mhh_pres_slab_plan_create(const mhh_grid* g, const void* host_dz, const void* host_dzhi, const void* host_rhoref, const void* host_rhorefh,
                                      mhh_pres_slab_plan** out)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(out && host_dz && host_dzhi && host_rhoref && host_rhorefh, "null pointer");
    MHH_REQUIRE(g->npy >= 1 && g->jtot % g->npy == 0 && g->jmax == g->jtot / g->npy && g->imax == g->itot, "slab grid: jmax = jtot/npy, imax = itot");
    MHH_REQUIRE(g->mpicoordy >= 0 && g->mpicoordy < g->npy, "mpicoordy");
    MHH_REQUIRE(g->jtot > 1 && g->kgc >= 1 && g->igc >= 1 && g->jgc >= 1, "pres_2 slab needs a 3-D grid with 1 ghost cell");
    mhh_pres_slab_plan* P = new mhh_pres_slab_plan();
    P->dtype = g->dtype; P->itot = g->itot; P->jtot = g->jtot; P->ktot = g->ktot; P->jmax = g->jmax; P->npy = g->npy; P->rank = g->mpicoordy;
    P->nxh = g->itot/2 + 1; P->nxb = (P->nxh + g->npy - 1) / g->npy; P->esz = (g->dtype == MHH_F64) ? 8 : 4;
    int e = (g->dtype == MHH_F64) ? slab_tables<double>(P, g, host_dz, host_dzhi, host_rhoref, host_rhorefh)
                                  : slab_tables<float>(P, g, host_dz, host_dzhi, host_rhoref, host_rhorefh);
    const size_t nreal = (size_t)g->itot*g->jmax*g->ktot, nx = (size_t)P->nxh*g->jmax*g->ktot, ny = (size_t)P->nxb*g->jtot*g->ktot;
    auto alloc = [&](void** p, size_t bytes) { if (e) return; hipError_t h = hipMalloc(p, bytes); if (h != hipSuccess) { set_error("hipMalloc: %s", hipGetErrorString(h)); e = MHH_ENOMEM; } };
    alloc(&P->packed, nreal*P->esz); alloc(&P->specx, nx*2*P->esz); alloc(&P->specy, ny*2*P->esz); alloc(&P->work, 2*ny*P->esz);          // pivots w2 and eliminated upper diagonal w3 of every column, factored below
    if (!e)
    {
        fft_acquire();
        const rocfft_array_type R = rocfft_array_type_real, H = rocfft_array_type_hermitian_interleaved, Cx = rocfft_array_type_complex_interleaved;
        const size_t bx = (size_t)g->jmax*g->ktot, by = (size_t)P->nxb*g->ktot;
        e = plan1d(&P->fx, rocfft_transform_type_real_forward, rocfft_placement_notinplace, g->dtype, g->itot, bx, R, H, g->itot, P->nxh, &P->wbs);
        if (!e) e = plan1d(&P->bx, rocfft_transform_type_real_inverse, rocfft_placement_notinplace, g->dtype, g->itot, bx, H, R, P->nxh, g->itot, &P->wbs);
        if (!e) e = plan1d(&P->fy, rocfft_transform_type_complex_forward, rocfft_placement_inplace, g->dtype, g->jtot, by, Cx, Cx, g->jtot, g->jtot, &P->wbs);
        if (!e) e = plan1d(&P->by, rocfft_transform_type_complex_inverse, rocfft_placement_inplace, g->dtype, g->jtot, by, Cx, Cx, g->jtot, g->jtot, &P->wbs);
        if (!e && rocfft_execution_info_create(&P->info) != rocfft_status_success) { set_error("FFT error: execution_info_create"); e = MHH_EFFT; }
        if (!e && P->wbs)
        {
            alloc(&P->wb, P->wbs); P->wb_cap = P->wbs;
            if (!e && rocfft_execution_info_set_work_buffer(P->info, P->wb, P->wbs) != rocfft_status_success) { set_error("FFT error: set_work_buffer"); e = MHH_EFFT; }
        }
    }
    if (!e) e = slab_factor(P);
    if (!e && lds_slab_usable(g)) { e = lds_slab_twiddles(g, &P->tx_lds); if (!e) e = lds_slab_twiddles_y(g, &P->ty_lds); }      // the transforms in LDS
    if (e) { mhh_pres_slab_plan_destroy(P); return e; }
    *out = P;
    return MHH_OK;
}
MHH_API unsigned long long mhh_pres_slab_xbuf_elems(const mhh_pres_slab_plan* P)   // complex elements of one all-to-all buffer
{
    return (unsigned long long)P->npy * P->ktot * P->jmax * P->nxb;
}
MHH_API void* mhh_pres_slab_packed(mhh_pres_slab_plan* P) { return P->packed; }

// specx [k][jl][kx] -> xbuf [q][k][jl][kxl] (FWD) and back (the reverse direction drops the zero padding)
template<class TF, bool FWD>
__global__ void __launch_bounds__(256) xbuf_x_kernel(C2<TF>* __restrict__ specx, C2<TF>* __restrict__ xbuf, int nxh, int nxb, int jmax, int ktot, int npy)
{
    const int kx = blockIdx.x*256 + threadIdx.x;          // 0 .. npy*nxb-1
    const int jl = blockIdx.y, k = blockIdx.z;
    if (kx >= npy*nxb) return;
    const int q = kx / nxb, kxl = kx - q*nxb;
    const size_t xb = (((size_t)q*ktot + k)*jmax + jl)*nxb + kxl;
    const size_t sx = ((size_t)k*jmax + jl)*nxh + kx;
    if (FWD) xbuf[xb] = (kx < nxh) ? specx[sx] : C2<TF>{0, 0};
    else if (kx < nxh) specx[sx] = xbuf[xb];
}
// xbuf [r][k][jl][kxl] <-> specy [k][kxl][j = r*jmax + jl]: per (k, r) a 2-D transpose of a (jmax x nxb) matrix of complex
// numbers, done through an LDS tile of 64 (j) x 32 (kxl) so that both the global reads and the global writes are
// contiguous runs (512 B rows on the xbuf side, 1 KB rows on the specy side). The tile rows are padded by one element:
// a transposed ds_read_b128 then touches 16 different 4-bank groups per lane group (no conflicts).
template<class TF, bool FWD>
__global__ void __launch_bounds__(256) xbuf_y_kernel(C2<TF>* __restrict__ specy, C2<TF>* __restrict__ xbuf, int nxb, int jmax, int jtot, int ktot)
{
    constexpr int TJ = 64, TX = 32;
    __shared__ C2<TF> tile[TJ][TX+1];
    const int ntx = (nxb + TX-1)/TX, njl = (jmax + TJ-1)/TJ;
    int b = blockIdx.x;
    const int txi = b % ntx; b /= ntx;
    const int jli = b % njl; const int r = b / njl;
    const int k = blockIdx.y;
    const int kx0 = txi*TX, jl0 = jli*TJ;
    const int t = threadIdx.x;
    const size_t xb0 = ((size_t)r*ktot + k)*jmax*nxb;            // start of the (r,k) matrix [jl][kxl]
    const size_t sy0 = (size_t)k*nxb*jtot + (size_t)r*jmax;      // specy element (k, kxl=0, j=r*jmax)
    if (FWD)
    {
        // all eight loads of a thread are requested before the first is used (clamped addresses, no branch between them): as
        // load - store pairs under a condition they were eight memory round trips in a row
        C2<TF> v[TJ*TX/256];
#pragma unroll
        for (int p = 0; p < TJ*TX/256; ++p)                      // 8 passes: 8 rows x 32 columns each
        {
            const int row = p*8 + t/32, cx = t % 32;
            const int jl = jl0 + row, kxl = kx0 + cx;
            const bool in = (jl < jmax && kxl < nxb);
            v[p] = xbuf[xb0 + (in ? (size_t)jl*nxb + kxl : (size_t)0)];
        }
#pragma unroll
        for (int p = 0; p < TJ*TX/256; ++p)
        {
            const int row = p*8 + t/32, cx = t % 32;
            if (jl0 + row < jmax && kx0 + cx < nxb) tile[row][cx] = v[p];
        }
        __syncthreads();
        for (int p = 0; p < TJ*TX/256; ++p)                      // 8 passes: 4 columns x 64 rows each
        {
            const int cx = p*4 + t/64, row = t % 64;
            const int jl = jl0 + row, kxl = kx0 + cx;
            if (jl < jmax && kxl < nxb) specy[sy0 + (size_t)kxl*jtot + jl] = tile[row][cx];
        }
    }
    else
    {
        C2<TF> v[TJ*TX/256];
#pragma unroll
        for (int p = 0; p < TJ*TX/256; ++p)
        {
            const int cx = p*4 + t/64, row = t % 64;
            const int jl = jl0 + row, kxl = kx0 + cx;
            const bool in = (jl < jmax && kxl < nxb);
            v[p] = specy[sy0 + (in ? (size_t)kxl*jtot + jl : (size_t)0)];
        }
#pragma unroll
        for (int p = 0; p < TJ*TX/256; ++p)
        {
            const int cx = p*4 + t/64, row = t % 64;
            if (jl0 + row < jmax && kx0 + cx < nxb) tile[row][cx] = v[p];
        }
        __syncthreads();
        for (int p = 0; p < TJ*TX/256; ++p)
        {
            const int row = p*8 + t/32, cx = t % 32;
            const int jl = jl0 + row, kxl = kx0 + cx;
            if (jl < jmax && kxl < nxb) xbuf[xb0 + (size_t)jl*nxb + kxl] = tile[row][cx];
        }
    }
}

// Thomas solve of the slab stage (src/pres_2.cxx:289-330 matrix, :202-263 tdma) on the rank's (x-block, all y) columns.
// A rank has only nxb*jtot columns (33 x 512 at 512^3 on 8 GPUs: half a wave per SIMD), so the sweep is latency-bound
// and organised for that:
//   * the pivots w2 and the eliminated upper diagonal w3 depend on the grid and (kx, ky) only: computed ONCE at plan
//     creation (tdma_slab_factor_kernel, the reference's recurrence) and kept, 2 x nxb*jtot*kmax values;
//   * one thread per (ky, re|im) component -- twice the parallelism, lanes still read consecutive 8-byte words;
//   * the right-hand side and pivots of the next 16 levels are loaded while the 16 current levels run through the
//     recurrence (only a multiply-subtract and the division by the pivot are left on the dependent chain).
template<class TF>
__global__ void __launch_bounds__(64) tdma_slab_factor_kernel(TF* __restrict__ W2, TF* __restrict__ W3,
                                                              const TF* __restrict__ bmati, const TF* __restrict__ bmatj,
                                                              const TF* __restrict__ a, const TF* __restrict__ c, const TF* __restrict__ dz, const TF* __restrict__ rho,
                                                              int nxh, int nxb, int kx0, int jtot, int kmax)
{
    const int ky = blockIdx.x*64 + threadIdx.x, kxl = blockIdx.y;
    const int kx = kx0 + kxl;
    if (ky >= jtot || kx >= nxh) return;
    const size_t kk = (size_t)nxb*jtot, col = (size_t)kxl*jtot + ky;
    const TF bm = bmati[kx] + bmatj[ky];
    const bool mean = (kx == 0 && ky == 0);
    TF w2;
    {
        const TF dz2 = dz[0]*dz[0];
        TF b = dz2 * rho[0]*bm - (a[0]+c[0]);
        b += a[0];
        if (kmax == 1) { if (mean) b -= c[0]; else b += c[0]; }
        w2 = b;
        W2[col] = w2; W3[col] = TF(0);
    }
    for (int k=1; k<kmax; ++k)
    {
        const TF dz2 = dz[k]*dz[k];
        TF b = dz2 * rho[k]*bm - (a[k]+c[k]);
        if (k == kmax-1) { if (mean) b -= c[k]; else b += c[k]; }
        const TF w3 = c[k-1] / w2;
        w2 = b - a[k]*w3;
        W2[col + (size_t)k*kk] = w2; W3[col + (size_t)k*kk] = w3;
    }
}
template<class TF>
__global__ void __launch_bounds__(128) tdma_slab_kernel(TF* __restrict__ p, const TF* __restrict__ W2, const TF* __restrict__ W3,
                                                        const TF* __restrict__ a, const TF* __restrict__ dz,
                                                        int nxh, int nxb, int kx0, int jtot, int kmax)
{
#ifndef MHH_SLAB_U
#define MHH_SLAB_U 16          // levels in flight per thread (512^3 / 8: y-stage 0.44 ms with 8, 0.41 with 16, 0.40 with 32)
#endif
    constexpr int U = MHH_SLAB_U;
    const int t = blockIdx.x*128 + threadIdx.x;           // 0 .. 2*jtot-1
    const int ky = t >> 1, comp = t & 1, kxl = blockIdx.y;
    const int kx = kx0 + kxl;                              // swapped indices: this rank owns a block of x modes, all y modes
    if (ky >= jtot || kx >= nxh) return;
    const size_t kk = (size_t)nxb*jtot, col = (size_t)kxl*jtot + ky;
    TF* __restrict__ q = p + 2*col + comp;                 // component stream, stride 2*kk per level
    const TF* __restrict__ w2 = W2 + col; const TF* __restrict__ w3 = W3 + col;
    TF pp;
    {
        TF v = dz[0]*dz[0] * q[0];
        v /= w2[0];
        q[0] = v; pp = v;
    }
    // forward sweep, levels 1 .. kmax-1 in blocks of U with the next block's operands in flight
    TF qc[U], wc[U];
#pragma unroll
    for (int u=0; u<U; ++u) { const int k = 1 + u; const bool ok = k < kmax; qc[u] = ok ? q[2*(size_t)k*kk] : TF(0); wc[u] = ok ? w2[(size_t)k*kk] : TF(1); }
    for (int k0=1; k0<kmax; k0+=U)
    {
        TF qn[U], wn[U];
#pragma unroll
        for (int u=0; u<U; ++u) { const int k = k0 + U + u; const bool ok = k < kmax; qn[u] = ok ? q[2*(size_t)k*kk] : TF(0); wn[u] = ok ? w2[(size_t)k*kk] : TF(1); }
#pragma unroll
        for (int u=0; u<U; ++u)
        {
            const int k = k0 + u;
            if (k < kmax)
            {
                TF v = dz[k]*dz[k] * qc[u];
                v -= a[k]*pp;
                v /= wc[u];
                q[2*(size_t)k*kk] = v; pp = v;
            }
        }
#pragma unroll
        for (int u=0; u<U; ++u) { qc[u] = qn[u]; wc[u] = wn[u]; }
    }
    // backward sweep, levels kmax-2 .. 0
#pragma unroll
    for (int u=0; u<U; ++u) { const int k = kmax-2 - u; const bool ok = k >= 0; qc[u] = ok ? q[2*(size_t)k*kk] : TF(0); wc[u] = ok ? w3[(size_t)(k+1)*kk] : TF(0); }
    for (int k0=kmax-2; k0>=0; k0-=U)
    {
        TF qn[U], wn[U];
#pragma unroll
        for (int u=0; u<U; ++u) { const int k = k0 - U - u; const bool ok = k >= 0; qn[u] = ok ? q[2*(size_t)k*kk] : TF(0); wn[u] = ok ? w3[(size_t)(k+1)*kk] : TF(0); }
#pragma unroll
        for (int u=0; u<U; ++u)
        {
            const int k = k0 - u;
            if (k >= 0)
            {
                TF v = qc[u];
                v -= wc[u]*pp;
                q[2*(size_t)k*kk] = v; pp = v;
            }
        }
#pragma unroll
        for (int u=0; u<U; ++u) { qc[u] = qn[u]; wc[u] = wn[u]; }
    }
}

static int slab_factor(mhh_pres_slab_plan* P)
{
    dim3 gf((P->jtot + 63)/64, P->nxb);
    const size_t ny = (size_t)P->nxb*P->jtot*P->ktot;
    if (P->dtype == MHH_F64)
        hipLaunchKernelGGL(tdma_slab_factor_kernel<double>, gf, dim3(64), 0, 0, (double*)P->work, (double*)P->work + ny, cp<double>(P->bmati), cp<double>(P->bmatj),
                           cp<double>(P->a), cp<double>(P->c), cp<double>(P->dz), cp<double>(P->rhoref), P->nxh, P->nxb, P->rank*P->nxb, P->jtot, P->ktot);
    else
        hipLaunchKernelGGL(tdma_slab_factor_kernel<float>, gf, dim3(64), 0, 0, (float*)P->work, (float*)P->work + ny, cp<float>(P->bmati), cp<float>(P->bmatj),
                           cp<float>(P->a), cp<float>(P->c), cp<float>(P->dz), cp<float>(P->rhoref), P->nxh, P->nxb, P->rank*P->nxb, P->jtot, P->ktot);
    hipError_t h = hipGetLastError(); if (h == hipSuccess) h = hipStreamSynchronize(0);
    if (h != hipSuccess) { set_error("tdma_slab_factor: %s", hipGetErrorString(h)); return MHH_EHIP; }
    return MHH_OK;
}

// packed real [k][jl][i] -> ghosted p: interior rows + x halo (wrap) + bottom ghost level; the y halo is the caller's exchange
template<class TF>
__global__ void __launch_bounds__(256) unpack_slab_kernel(TF* __restrict__ p, const TF* __restrict__ packed, int itot, int jtot, int jmax, int kmax,
                                                          int igc, int jgc, int kgc, int icells, int jcells)
{
    const int i = blockIdx.x*256 + threadIdx.x;
    const int jl = blockIdx.y, kz = blockIdx.z;
    if (i >= icells) return;
    const int kd = (kz < kmax) ? kz + kgc : kgc - 1;
    const int ks = (kz < kmax) ? kz : 0;
    int is = (i - igc) % itot; if (is < 0) is += itot;
    const TF val = packed[(size_t)is + (size_t)jl*itot + (size_t)ks*itot*jmax] / jtot / itot;
    p[(size_t)i + (size_t)(jl + jgc)*icells + (size_t)kd*icells*jcells] = val;
}

// unpack and Pres_2::output in one pass on a slab (as unpack_out2_kernel of k_pres.hip): p of the rank's own rows, and on
// interior cells ut, wt and -- for every row but the southernmost, whose p[j-1] lives on the south neighbour -- vt, with the
// neighbours re-normalised from the packed solution (the stored values). The southernmost row of vt follows the one-row
// halo exchange of p (pres_out_south_row_kernel).
template<bool POW2, class TF>
__global__ void __launch_bounds__(256) unpack_out_slab_kernel(TF* __restrict__ p, const TF* __restrict__ packed,
                                                              TF* __restrict__ ut, TF* __restrict__ vt, TF* __restrict__ wt, const TF* __restrict__ dzhi,
                                                              TF dxi, TF dyi, int itot, int jtot, int jmax, int kmax, int igc, int jgc, int kgc, int icells, int jcells)
{
    const int i = blockIdx.x*256 + threadIdx.x;
    const int jl = blockIdx.y, kz = blockIdx.z;
    if (i >= icells) return;
    const int kd = (kz < kmax) ? kz + kgc : kgc - 1;
    const int ks = (kz < kmax) ? kz : 0;
    int is = (i - igc) % itot; if (is < 0) is += itot;
    const size_t ijm = (size_t)itot*jmax;
    const TF ri = TF(1)/TF(itot), rj = TF(1)/TF(jtot);
    const TF pc = fft_norm<POW2>(packed[(size_t)is + (size_t)jl*itot + (size_t)ks*ijm], itot, jtot, ri, rj);
    const size_t c = (size_t)i + (size_t)(jl + jgc)*icells + (size_t)kd*icells*jcells;
    p[c] = pc;
    if (kz < kmax && i >= igc && i < igc + itot)
    {
        const int iw = (is == 0) ? itot-1 : is-1;
        const TF pw = fft_norm<POW2>(packed[(size_t)iw + (size_t)jl*itot + (size_t)ks*ijm], itot, jtot, ri, rj);
        const TF pb = (ks == 0) ? pc : fft_norm<POW2>(packed[(size_t)is + (size_t)jl*itot + (size_t)(ks-1)*ijm], itot, jtot, ri, rj);
        ut[c] -= (pc - pw) * dxi;
        if (jl > 0)
        {
            const TF ps = fft_norm<POW2>(packed[(size_t)is + (size_t)(jl-1)*itot + (size_t)ks*ijm], itot, jtot, ri, rj);
            vt[c] -= (pc - ps) * dyi;
        }
        wt[c] -= (pc - pb) * dzhi[kd];
    }
}
template<class TF>
__global__ void __launch_bounds__(256) pres_out_south_row_kernel(TF* __restrict__ vt, const TF* __restrict__ p, TF dyi, int istart, int iend, int jstart, int kstart, int icells, int ijcells)
{
    const int i = istart + blockIdx.x*256 + threadIdx.x;
    if (i >= iend) return;
    const size_t c = (size_t)i + (size_t)jstart*icells + (size_t)(kstart + blockIdx.y)*ijcells;
    vt[c] -= (p[c] - p[c-icells]) * dyi;
}

static int slab_match(const mhh_pres_slab_plan* P, const mhh_grid* g)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(P && P->dtype == g->dtype && P->itot == g->itot && P->jtot == g->jtot && P->ktot == g->ktot && P->jmax == g->jmax && P->npy == g->npy && P->rank == g->mpicoordy, "plan/grid mismatch");
    return MHH_OK;
}

// stage 1: x transform of the packed divergence + pack for the forward all-to-all
MHH_API int mhh_pres_fwd_x_pack(mhh_pres_slab_plan* P, const mhh_grid* g, void* p_packed, void* sendbuf, void* stream)
{
    if (int e = slab_match(P, g)) return e;
    MHH_REQUIRE(sendbuf != nullptr, "sendbuf");
    if (!p_packed) p_packed = P->packed;
    hipStream_t st = as_stream(stream);
    MHH_FFT_TRY(rocfft_execution_info_set_stream(P->info, st));
    void* in[1] = {p_packed}; void* out[1] = {P->specx};
    MHH_FFT_TRY(rocfft_execute(P->fx, in, out, P->info));
    dim3 grid((P->npy*P->nxb + 255)/256, P->jmax, P->ktot);
    if (g->dtype == MHH_F64) hipLaunchKernelGGL((xbuf_x_kernel<double, true>), grid, dim3(256), 0, st, (C2<double>*)P->specx, (C2<double>*)sendbuf, P->nxh, P->nxb, P->jmax, P->ktot, P->npy);
    else                     hipLaunchKernelGGL((xbuf_x_kernel<float, true>), grid, dim3(256), 0, st, (C2<float>*)P->specx, (C2<float>*)sendbuf, P->nxh, P->nxb, P->jmax, P->ktot, P->npy);
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
// stage 2: after the forward all-to-all: y transform, tridiagonal solves, inverse y transform, pack for the way back
MHH_API int mhh_pres_fwd_y_solve_bwd_y(mhh_pres_slab_plan* P, const mhh_grid* g, void* recvbuf, void* sendbuf, void* stream)
{
    if (int e = slab_match(P, g)) return e;
    MHH_REQUIRE(recvbuf && sendbuf, "buffers");
    hipStream_t st = as_stream(stream);
    MHH_FFT_TRY(rocfft_execution_info_set_stream(P->info, st));
    dim3 gy((unsigned)(((P->nxb + 31)/32) * ((P->jmax + 63)/64) * P->npy), P->ktot);
    dim3 gs((2*P->jtot + 127)/128, P->nxb);
    void* io[1] = {P->specy};
    if (g->dtype == MHH_F64)
    {
        hipLaunchKernelGGL((xbuf_y_kernel<double, true>), gy, dim3(256), 0, st, (C2<double>*)P->specy, (C2<double>*)recvbuf, P->nxb, P->jmax, P->jtot, P->ktot);
        MHH_LAUNCH_CHECK();
        MHH_FFT_TRY(rocfft_execute(P->fy, io, nullptr, P->info));
        hipLaunchKernelGGL(tdma_slab_kernel<double>, gs, dim3(128), 0, st, (double*)P->specy, cp<double>(P->work), cp<double>(P->work) + (size_t)P->nxb*P->jtot*P->ktot,
                           cp<double>(P->a), cp<double>(P->dz), P->nxh, P->nxb, P->rank*P->nxb, P->jtot, P->ktot);
        MHH_LAUNCH_CHECK();
        MHH_FFT_TRY(rocfft_execute(P->by, io, nullptr, P->info));
        hipLaunchKernelGGL((xbuf_y_kernel<double, false>), gy, dim3(256), 0, st, (C2<double>*)P->specy, (C2<double>*)sendbuf, P->nxb, P->jmax, P->jtot, P->ktot);
    }
    else
    {
        hipLaunchKernelGGL((xbuf_y_kernel<float, true>), gy, dim3(256), 0, st, (C2<float>*)P->specy, (C2<float>*)recvbuf, P->nxb, P->jmax, P->jtot, P->ktot);
        MHH_LAUNCH_CHECK();
        MHH_FFT_TRY(rocfft_execute(P->fy, io, nullptr, P->info));
        hipLaunchKernelGGL(tdma_slab_kernel<float>, gs, dim3(128), 0, st, (float*)P->specy, cp<float>(P->work), cp<float>(P->work) + (size_t)P->nxb*P->jtot*P->ktot,
                           cp<float>(P->a), cp<float>(P->dz), P->nxh, P->nxb, P->rank*P->nxb, P->jtot, P->ktot);
        MHH_LAUNCH_CHECK();
        MHH_FFT_TRY(rocfft_execute(P->by, io, nullptr, P->info));
        hipLaunchKernelGGL((xbuf_y_kernel<float, false>), gy, dim3(256), 0, st, (C2<float>*)P->specy, (C2<float>*)sendbuf, P->nxb, P->jmax, P->jtot, P->ktot);
    }
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
// stage 3: after the backward all-to-all: inverse x transform, normalise, write p (interior rows, x halo, bottom ghost level)
MHH_API int mhh_pres_bwd_x_unpack(mhh_pres_slab_plan* P, const mhh_grid* g, void* recvbuf, const mhh_fields* f, void* stream)
{
    if (int e = slab_match(P, g)) return e;
    MHH_REQUIRE(recvbuf && f && f->p, "buffers");
    hipStream_t st = as_stream(stream);
    MHH_FFT_TRY(rocfft_execution_info_set_stream(P->info, st));
    dim3 grid((P->npy*P->nxb + 255)/256, P->jmax, P->ktot);
    dim3 ug((g->icells + 255)/256, g->jmax, g->kmax + 1);
    void* in[1] = {P->specx}; void* out[1] = {P->packed};
    if (g->dtype == MHH_F64)
    {
        hipLaunchKernelGGL((xbuf_x_kernel<double, false>), grid, dim3(256), 0, st, (C2<double>*)P->specx, (C2<double>*)recvbuf, P->nxh, P->nxb, P->jmax, P->ktot, P->npy);
        MHH_LAUNCH_CHECK();
        MHH_FFT_TRY(rocfft_execute(P->bx, in, out, P->info));
        hipLaunchKernelGGL(unpack_slab_kernel<double>, ug, dim3(256), 0, st, mp<double>(f->p), cp<double>(P->packed), g->itot, g->jtot, g->jmax, g->kmax, g->igc, g->jgc, g->kgc, g->icells, g->jcells);
    }
    else
    {
        hipLaunchKernelGGL((xbuf_x_kernel<float, false>), grid, dim3(256), 0, st, (C2<float>*)P->specx, (C2<float>*)recvbuf, P->nxh, P->nxb, P->jmax, P->ktot, P->npy);
        MHH_LAUNCH_CHECK();
        MHH_FFT_TRY(rocfft_execute(P->bx, in, out, P->info));
        hipLaunchKernelGGL(unpack_slab_kernel<float>, ug, dim3(256), 0, st, mp<float>(f->p), cp<float>(P->packed), g->itot, g->jtot, g->jmax, g->kmax, g->igc, g->jgc, g->kgc, g->icells, g->jcells);
    }
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
// stage 3, fused form: inverse x transform, then unpack + Pres_2::output in one kernel for everything but vt on the southernmost
// row; the caller exchanges the one-row halo of p and finishes with mhh_pres_output_south_row.
MHH_API int mhh_pres_bwd_x_unpack_output(mhh_pres_slab_plan* P, const mhh_grid* g, void* recvbuf, const mhh_fields* f, void* stream)
{
    if (int e = slab_match(P, g)) return e;
    MHH_REQUIRE(recvbuf && f && f->p && f->ut && f->vt && f->wt, "buffers");
    hipStream_t st = as_stream(stream);
    MHH_FFT_TRY(rocfft_execution_info_set_stream(P->info, st));
    dim3 grid((P->npy*P->nxb + 255)/256, P->jmax, P->ktot);
    dim3 ug((g->icells + 255)/256, g->jmax, g->kmax + 1);
    void* in[1] = {P->specx}; void* out[1] = {P->packed};
    const bool pow2 = is_pow2(g->itot) && is_pow2(g->jtot);
#define CALL(TF) [&]{ const GridDev<TF> gd = make_grid<TF>(g); \
        hipLaunchKernelGGL((xbuf_x_kernel<TF, false>), grid, dim3(256), 0, st, (C2<TF>*)P->specx, (C2<TF>*)recvbuf, P->nxh, P->nxb, P->jmax, P->ktot, P->npy); \
        if (hipGetLastError() != hipSuccess) return (int)MHH_EHIP; \
        if (rocfft_execute(P->bx, in, out, P->info) != rocfft_status_success) return (int)MHH_EFFT; \
        if (pow2) hipLaunchKernelGGL((unpack_out_slab_kernel<true, TF>), ug, dim3(256), 0, st, mp<TF>(f->p), cp<TF>(P->packed), mp<TF>(f->ut), mp<TF>(f->vt), mp<TF>(f->wt), gd.dzhi, \
                           gd.dxi_t, gd.dyi_t, g->itot, g->jtot, g->jmax, g->kmax, g->igc, g->jgc, g->kgc, g->icells, g->jcells); \
        else hipLaunchKernelGGL((unpack_out_slab_kernel<false, TF>), ug, dim3(256), 0, st, mp<TF>(f->p), cp<TF>(P->packed), mp<TF>(f->ut), mp<TF>(f->vt), mp<TF>(f->wt), gd.dzhi, \
                           gd.dxi_t, gd.dyi_t, g->itot, g->jtot, g->jmax, g->kmax, g->igc, g->jgc, g->kgc, g->icells, g->jcells); return (int)MHH_OK; }()
    if (int e = MHH_DISPATCH(g, CALL)) { set_error("pres_bwd_x_unpack_output: launch / FFT error"); return e; }
#undef CALL
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
MHH_API int mhh_pres_output_south_row(const mhh_grid* g, const mhh_fields* f, void* stream)
{
    if (int e = check_grid(g)) return e;
    MHH_REQUIRE(f && f->p && f->vt && g->jgc >= 1, "null field");
    dim3 grid((g->imax + 255)/256, g->kmax);
#define CALL(TF) [&]{ const GridDev<TF> gd = make_grid<TF>(g); \
        hipLaunchKernelGGL(pres_out_south_row_kernel<TF>, grid, dim3(256), 0, as_stream(stream), mp<TF>(f->vt), cp<TF>(f->p), gd.dyi_t, g->istart, g->iend, g->jstart, g->kstart, g->icells, g->ijcells); return (int)MHH_OK; }()
    if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}


// =======================================================================================================
// The same solve in k-slices (src/fft.cxx:451-583 transforms the planes of a field one k-slice at a time too): the buffers of
// the all-to-alls are laid out [slice][peer][k in slice][jl][kxl], so that slice c is ONE equal-split all-to-all of its own,
// which the host issues on a second stream while slice c+1 is being transformed (microhh_amd/model.py: HotPath.pres). Every
// plane goes through the same transform kernels as in the unsliced call; the Thomas sweeps need all slices.
// =======================================================================================================
MHH_API int mhh_pres_slab_set_chunks(mhh_pres_slab_plan* P, int nchunks)
{
    MHH_REQUIRE(P && nchunks >= 1 && P->ktot % nchunks == 0, "the number of k-slices must divide ktot");
    for (rocfft_plan* p : {&P->cfx, &P->cbx, &P->cfy, &P->cby}) if (*p) { rocfft_plan_destroy(*p); *p = nullptr; }
    P->nchunks = nchunks;
    if (nchunks == 1) return MHH_OK;
    const int kc = P->ktot / nchunks;
    const rocfft_array_type R = rocfft_array_type_real, H = rocfft_array_type_hermitian_interleaved, Cx = rocfft_array_type_complex_interleaved;
    const size_t bx = (size_t)P->jmax*kc, by = (size_t)P->nxb*kc;
    size_t wbs = P->wbs;
    int e = plan1d(&P->cfx, rocfft_transform_type_real_forward, rocfft_placement_notinplace, P->dtype, P->itot, bx, R, H, P->itot, P->nxh, &wbs);
    if (!e) e = plan1d(&P->cbx, rocfft_transform_type_real_inverse, rocfft_placement_notinplace, P->dtype, P->itot, bx, H, R, P->nxh, P->itot, &wbs);
    if (!e) e = plan1d(&P->cfy, rocfft_transform_type_complex_forward, rocfft_placement_inplace, P->dtype, P->jtot, by, Cx, Cx, P->jtot, P->jtot, &wbs);
    if (!e) e = plan1d(&P->cby, rocfft_transform_type_complex_inverse, rocfft_placement_inplace, P->dtype, P->jtot, by, Cx, Cx, P->jtot, P->jtot, &wbs);
    if (!e && wbs > P->wb_cap)
    {
        MHH_HIP_TRY(hipStreamSynchronize(0));          // (the work buffer may still be in use by a solve in flight on the default stream)
        if (P->wb) (void)hipFree(P->wb);
        MHH_HIP_TRY(hipMalloc(&P->wb, wbs)); P->wb_cap = wbs;
        MHH_FFT_TRY(rocfft_execution_info_set_work_buffer(P->info, P->wb, wbs));
    }
    P->wbs = wbs;
    return e;
}
MHH_API int mhh_pres_slab_chunks(const mhh_pres_slab_plan* P) { return P ? P->nchunks : 0; }

namespace
{
struct Slice { int kc; size_t xseg, sx, sy, pk; };      // levels per slice; element offsets of slice c in xbuf / specx / specy / packed
inline Slice slice_of(const mhh_pres_slab_plan* P, int c)
{
    const int kc = P->ktot / P->nchunks;
    return Slice{kc, (size_t)c*P->npy*kc*P->jmax*P->nxb, (size_t)c*kc*P->jmax*P->nxh, (size_t)c*kc*P->nxb*P->jtot, (size_t)c*kc*P->jmax*P->itot};
}
inline int chunk_ok(const mhh_pres_slab_plan* P, const mhh_grid* g, int c)
{
    if (int e = slab_match(P, g)) return e;
    MHH_REQUIRE(P->nchunks > 1 && c >= 0 && c < P->nchunks, "k-slice index (call mhh_pres_slab_set_chunks first)");
    return MHH_OK;
}
}
// slice c: x transform of the packed divergence + pack into segment c of sendbuf
MHH_API int mhh_pres_fwd_x_pack_chunk(mhh_pres_slab_plan* P, const mhh_grid* g, void* p_packed, void* sendbuf, int c, void* stream)
{
    if (int e = chunk_ok(P, g, c)) return e;
    MHH_REQUIRE(sendbuf != nullptr, "sendbuf");
    if (!p_packed) p_packed = P->packed;
    const Slice sl = slice_of(P, c);
    hipStream_t st = as_stream(stream);
    MHH_FFT_TRY(rocfft_execution_info_set_stream(P->info, st));
    void* in[1] = {static_cast<char*>(p_packed) + sl.pk*P->esz}; void* out[1] = {static_cast<char*>(P->specx) + sl.sx*2*P->esz};
    MHH_FFT_TRY(rocfft_execute(P->cfx, in, out, P->info));
    dim3 grid((P->npy*P->nxb + 255)/256, P->jmax, sl.kc);
    if (g->dtype == MHH_F64) hipLaunchKernelGGL((xbuf_x_kernel<double, true>), grid, dim3(256), 0, st, (C2<double>*)P->specx + sl.sx, (C2<double>*)sendbuf + sl.xseg, P->nxh, P->nxb, P->jmax, sl.kc, P->npy);
    else                     hipLaunchKernelGGL((xbuf_x_kernel<float, true>), grid, dim3(256), 0, st, (C2<float>*)P->specx + sl.sx, (C2<float>*)sendbuf + sl.xseg, P->nxh, P->nxb, P->jmax, sl.kc, P->npy);
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
// slice c, after its forward all-to-all: reorder into [k][kxl][j] and forward y transform
MHH_API int mhh_pres_fwd_y_chunk(mhh_pres_slab_plan* P, const mhh_grid* g, void* recvbuf, int c, void* stream)
{
    if (int e = chunk_ok(P, g, c)) return e;
    MHH_REQUIRE(recvbuf != nullptr, "recvbuf");
    const Slice sl = slice_of(P, c);
    hipStream_t st = as_stream(stream);
    MHH_FFT_TRY(rocfft_execution_info_set_stream(P->info, st));
    dim3 gy((unsigned)(((P->nxb + 31)/32) * ((P->jmax + 63)/64) * P->npy), sl.kc);
    void* io[1] = {static_cast<char*>(P->specy) + sl.sy*2*P->esz};
    if (g->dtype == MHH_F64) hipLaunchKernelGGL((xbuf_y_kernel<double, true>), gy, dim3(256), 0, st, (C2<double>*)P->specy + sl.sy, (C2<double>*)recvbuf + sl.xseg, P->nxb, P->jmax, P->jtot, sl.kc);
    else                     hipLaunchKernelGGL((xbuf_y_kernel<float, true>), gy, dim3(256), 0, st, (C2<float>*)P->specy + sl.sy, (C2<float>*)recvbuf + sl.xseg, P->nxb, P->jmax, P->jtot, sl.kc);
    MHH_LAUNCH_CHECK();
    MHH_FFT_TRY(rocfft_execute(P->cfy, io, nullptr, P->info));
    return MHH_OK;
}
// all slices in: the tridiagonal solves over k
MHH_API int mhh_pres_solve_y(mhh_pres_slab_plan* P, const mhh_grid* g, void* stream)
{
    if (int e = slab_match(P, g)) return e;
    hipStream_t st = as_stream(stream);
    dim3 gs((2*P->jtot + 127)/128, P->nxb);
    if (g->dtype == MHH_F64)
        hipLaunchKernelGGL(tdma_slab_kernel<double>, gs, dim3(128), 0, st, (double*)P->specy, cp<double>(P->work), cp<double>(P->work) + (size_t)P->nxb*P->jtot*P->ktot,
                           cp<double>(P->a), cp<double>(P->dz), P->nxh, P->nxb, P->rank*P->nxb, P->jtot, P->ktot);
    else
        hipLaunchKernelGGL(tdma_slab_kernel<float>, gs, dim3(128), 0, st, (float*)P->specy, cp<float>(P->work), cp<float>(P->work) + (size_t)P->nxb*P->jtot*P->ktot,
                           cp<float>(P->a), cp<float>(P->dz), P->nxh, P->nxb, P->rank*P->nxb, P->jtot, P->ktot);
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
// slice c: inverse y transform + reorder into segment c of sendbuf
MHH_API int mhh_pres_bwd_y_chunk(mhh_pres_slab_plan* P, const mhh_grid* g, void* sendbuf, int c, void* stream)
{
    if (int e = chunk_ok(P, g, c)) return e;
    MHH_REQUIRE(sendbuf != nullptr, "sendbuf");
    const Slice sl = slice_of(P, c);
    hipStream_t st = as_stream(stream);
    MHH_FFT_TRY(rocfft_execution_info_set_stream(P->info, st));
    dim3 gy((unsigned)(((P->nxb + 31)/32) * ((P->jmax + 63)/64) * P->npy), sl.kc);
    void* io[1] = {static_cast<char*>(P->specy) + sl.sy*2*P->esz};
    MHH_FFT_TRY(rocfft_execute(P->cby, io, nullptr, P->info));
    if (g->dtype == MHH_F64) hipLaunchKernelGGL((xbuf_y_kernel<double, false>), gy, dim3(256), 0, st, (C2<double>*)P->specy + sl.sy, (C2<double>*)sendbuf + sl.xseg, P->nxb, P->jmax, P->jtot, sl.kc);
    else                     hipLaunchKernelGGL((xbuf_y_kernel<float, false>), gy, dim3(256), 0, st, (C2<float>*)P->specy + sl.sy, (C2<float>*)sendbuf + sl.xseg, P->nxb, P->jmax, P->jtot, sl.kc);
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
// slice c, after its backward all-to-all: back into [k][jl][kx] and inverse x transform into the packed solution
MHH_API int mhh_pres_bwd_x_chunk(mhh_pres_slab_plan* P, const mhh_grid* g, void* recvbuf, int c, void* stream)
{
    if (int e = chunk_ok(P, g, c)) return e;
    MHH_REQUIRE(recvbuf != nullptr, "recvbuf");
    const Slice sl = slice_of(P, c);
    hipStream_t st = as_stream(stream);
    MHH_FFT_TRY(rocfft_execution_info_set_stream(P->info, st));
    dim3 grid((P->npy*P->nxb + 255)/256, P->jmax, sl.kc);
    if (g->dtype == MHH_F64) hipLaunchKernelGGL((xbuf_x_kernel<double, false>), grid, dim3(256), 0, st, (C2<double>*)P->specx + sl.sx, (C2<double>*)recvbuf + sl.xseg, P->nxh, P->nxb, P->jmax, sl.kc, P->npy);
    else                     hipLaunchKernelGGL((xbuf_x_kernel<float, false>), grid, dim3(256), 0, st, (C2<float>*)P->specx + sl.sx, (C2<float>*)recvbuf + sl.xseg, P->nxh, P->nxb, P->jmax, sl.kc, P->npy);
    MHH_LAUNCH_CHECK();
    void* in[1] = {static_cast<char*>(P->specx) + sl.sx*2*P->esz}; void* out[1] = {static_cast<char*>(P->packed) + sl.pk*P->esz};
    MHH_FFT_TRY(rocfft_execute(P->cbx, in, out, P->info));
    return MHH_OK;
}
// ---- the x stages with the transforms in LDS (pres_lds.h): input + x transform WRITING the send buffer of the x -> y transpose,
// x transform + p + output READING the receive buffer of the y -> x transpose -- two kernels and four array passes per rank where
// the staged form has input | x r2c | pack and unpack | x c2r | unpack + output (six kernels, eleven passes). The y stage between
// the transposes is unchanged. c = k-slice (0 with unsliced transposes): the kernels work the levels of that slice only.
MHH_API int mhh_pres_slab_has_lds(const mhh_pres_slab_plan* P)
{
    const char* e = getenv("MHH_PRES_SLAB_LDS");          // "0": the staged x stages (A/B runs, tests of both forms)
    return (P && P->tx_lds && !(e && !strcmp(e, "0"))) ? 1 : 0;
}
MHH_API int mhh_pres_slab_lds_fwd(mhh_pres_slab_plan* P, const mhh_grid* g, const mhh_fields* f, double dt, void* sendbuf, int c, void* stream)
{
    if (int e = slab_match(P, g)) return e;
    MHH_REQUIRE(P->tx_lds != nullptr, "this plan has no LDS form of the x stages (power-of-two itot, jmax a multiple of 8)");
    MHH_REQUIRE(f && f->u && f->v && f->w && f->ut && f->vt && f->wt && f->rhoref && f->rhorefh && sendbuf, "null field");
    MHH_REQUIRE(dt > 0. && c >= 0 && c < P->nchunks, "dt, k-slice");
    const int ks = P->ktot / P->nchunks;
    return lds_slab_stage_in(g, f, dt, sendbuf, P->tx_lds, P->nxb, P->npy, ks, c*ks, (c+1)*ks, as_stream(stream));
}
MHH_API int mhh_pres_slab_lds_bwd(mhh_pres_slab_plan* P, const mhh_grid* g, const void* recvbuf, const mhh_fields* f, int c, void* stream)
{
    if (int e = slab_match(P, g)) return e;
    MHH_REQUIRE(P->tx_lds != nullptr, "this plan has no LDS form of the x stages (power-of-two itot, jmax a multiple of 8)");
    MHH_REQUIRE(f && f->p && f->ut && f->vt && f->wt && recvbuf, "null field");
    MHH_REQUIRE(c >= 0 && c < P->nchunks, "k-slice");
    const int ks = P->ktot / P->nchunks;
    return lds_slab_stage_out(g, f, recvbuf, P->tx_lds, P->nxb, P->npy, ks, c*ks, (c+1)*ks, as_stream(stream));
}
// The y stage of the LDS form: the transform along y of k-slice c from the receive buffer into the plan's [k][kxl][ky] array
// (mhh_pres_slab_lds_fwd_y), the Thomas sweeps over all levels (mhh_pres_solve_y), the transform back into the send buffer
// (mhh_pres_slab_lds_bwd_y). The buffers of this form are laid out [slice][peer][k][kxl][row] (rows fastest), NOT as the staged form's.
MHH_API int mhh_pres_slab_lds_fwd_y(mhh_pres_slab_plan* P, const mhh_grid* g, void* recvbuf, int c, void* stream)
{
    if (int e = slab_match(P, g)) return e;
    MHH_REQUIRE(P->ty_lds != nullptr && recvbuf && c >= 0 && c < P->nchunks, "LDS form, buffer, k-slice");
    const int ks = P->ktot / P->nchunks;
    return lds_slab_yfft(g, true, recvbuf, P->specy, P->ty_lds, P->nxb, P->npy, ks, c*ks, (c+1)*ks, as_stream(stream));
}
MHH_API int mhh_pres_slab_lds_bwd_y(mhh_pres_slab_plan* P, const mhh_grid* g, void* sendbuf, int c, void* stream)
{
    if (int e = slab_match(P, g)) return e;
    MHH_REQUIRE(P->ty_lds != nullptr && sendbuf && c >= 0 && c < P->nchunks, "LDS form, buffer, k-slice");
    const int ks = P->ktot / P->nchunks;
    return lds_slab_yfft(g, false, sendbuf, P->specy, P->ty_lds, P->nxb, P->npy, ks, c*ks, (c+1)*ks, as_stream(stream));
}
// all slices back: unpack + Pres_2::output in one kernel (the tail of mhh_pres_bwd_x_unpack_output)
MHH_API int mhh_pres_unpack_output_slab(mhh_pres_slab_plan* P, const mhh_grid* g, const mhh_fields* f, void* stream)
{
    if (int e = slab_match(P, g)) return e;
    MHH_REQUIRE(f && f->p && f->ut && f->vt && f->wt, "buffers");
    hipStream_t st = as_stream(stream);
    dim3 ug((g->icells + 255)/256, g->jmax, g->kmax + 1);
    const bool pow2 = is_pow2(g->itot) && is_pow2(g->jtot);
#define CALL(TF) [&]{ const GridDev<TF> gd = make_grid<TF>(g); \
        if (pow2) hipLaunchKernelGGL((unpack_out_slab_kernel<true, TF>), ug, dim3(256), 0, st, mp<TF>(f->p), cp<TF>(P->packed), mp<TF>(f->ut), mp<TF>(f->vt), mp<TF>(f->wt), gd.dzhi, \
                           gd.dxi_t, gd.dyi_t, g->itot, g->jtot, g->jmax, g->kmax, g->igc, g->jgc, g->kgc, g->icells, g->jcells); \
        else hipLaunchKernelGGL((unpack_out_slab_kernel<false, TF>), ug, dim3(256), 0, st, mp<TF>(f->p), cp<TF>(P->packed), mp<TF>(f->ut), mp<TF>(f->vt), mp<TF>(f->wt), gd.dzhi, \
                           gd.dxi_t, gd.dyi_t, g->itot, g->jtot, g->jmax, g->kmax, g->igc, g->jgc, g->kgc, g->icells, g->jcells); return (int)MHH_OK; }()
    if (int e = MHH_DISPATCH(g, CALL)) return e;
#undef CALL
    MHH_LAUNCH_CHECK();
    return MHH_OK;
}
