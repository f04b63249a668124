// Probe: does this rocFFT honour load/store callbacks on the batched 2-D real-to-complex / complex-to-real plans the
// pressure solver uses, and what do they cost? (experiment, not part of the library)
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define CHK(x) do { auto e_ = (x); if (e_ != 0) { std::printf("FAIL %s -> %d\n", #x, (int)e_); return 1; } } while (0)

struct CbData { const double* a; const double* b; double scale; };
__device__ double load_cb(double* data, size_t offset, void* cbdata, void*)
{
    const CbData* d = static_cast<const CbData*>(cbdata);
    return d->scale * (d->a[offset] + d->b[offset]);
}
__device__ auto load_cb_ptr = load_cb;
__device__ void store_cb(double* data, size_t offset, double v, void* cbdata, void*)
{
    const CbData* d = static_cast<const CbData*>(cbdata);
    data[offset] = v * d->scale;
}
__device__ auto store_cb_ptr = store_cb;

int main()
{
    const size_t nx = 512, ny = 512, nb = 128, nxh = nx/2 + 1;
    const size_t nreal = nx*ny*nb, ncx = nxh*ny*nb;
    std::vector<double> ha(nreal), hb(nreal);
    for (size_t n=0; n<nreal; ++n) { ha[n] = std::sin(0.001*n); hb[n] = std::cos(0.0007*n); }
    double *a, *b, *sum, *out1, *out2; double2 *spec1, *spec2;
    CHK(hipMalloc(&a, nreal*8)); CHK(hipMalloc(&b, nreal*8)); CHK(hipMalloc(&sum, nreal*8)); CHK(hipMalloc(&out1, nreal*8)); CHK(hipMalloc(&out2, nreal*8));
    CHK(hipMalloc(&spec1, ncx*16)); CHK(hipMalloc(&spec2, ncx*16));
    CHK(hipMemcpy(a, ha.data(), nreal*8, hipMemcpyHostToDevice)); CHK(hipMemcpy(b, hb.data(), nreal*8, hipMemcpyHostToDevice));
    for (size_t n=0; n<nreal; ++n) ha[n] = 0.5*(ha[n] + hb[n]);
    CHK(hipMemcpy(sum, ha.data(), nreal*8, hipMemcpyHostToDevice));
    rocfft_setup();
    size_t lengths[2] = {nx, ny}, rstr[2] = {1, nx}, cstr[2] = {1, nxh}, off[2] = {0, 0};
    rocfft_plan_description df, dbk; rocfft_plan pf, pb;
    CHK(rocfft_plan_description_create(&df));
    CHK(rocfft_plan_description_set_data_layout(df, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, off, off, 2, rstr, nx*ny, 2, cstr, nxh*ny));
    CHK(rocfft_plan_create(&pf, rocfft_placement_notinplace, rocfft_transform_type_real_forward, rocfft_precision_double, 2, lengths, nb, df));
    CHK(rocfft_plan_description_create(&dbk));
    CHK(rocfft_plan_description_set_data_layout(dbk, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, off, off, 2, cstr, nxh*ny, 2, rstr, nx*ny));
    CHK(rocfft_plan_create(&pb, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, rocfft_precision_double, 2, lengths, nb, dbk));
    rocfft_execution_info i0, i1, i2, i3;
    CHK(rocfft_execution_info_create(&i0)); CHK(rocfft_execution_info_create(&i1)); CHK(rocfft_execution_info_create(&i2)); CHK(rocfft_execution_info_create(&i3));
    size_t wf = 0, wb = 0; rocfft_plan_get_work_buffer_size(pf, &wf); rocfft_plan_get_work_buffer_size(pb, &wb);
    void* wbuf = nullptr; size_t wmax = wf > wb ? wf : wb;
    if (wmax) { CHK(hipMalloc(&wbuf, wmax)); for (auto i : {i0, i1, i2, i3}) CHK(rocfft_execution_info_set_work_buffer(i, wbuf, wmax)); }
    CbData hd{a, b, 0.5}; CbData* dd; CHK(hipMalloc(&dd, sizeof(CbData))); CHK(hipMemcpy(dd, &hd, sizeof(CbData), hipMemcpyHostToDevice));
    CbData hs{nullptr, nullptr, 1.0/(nx*ny)}; CbData* ds; CHK(hipMalloc(&ds, sizeof(CbData))); CHK(hipMemcpy(ds, &hs, sizeof(CbData), hipMemcpyHostToDevice));
    void* lfn; void* sfn;
    CHK(hipMemcpyFromSymbol(&lfn, HIP_SYMBOL(load_cb_ptr), sizeof(void*))); CHK(hipMemcpyFromSymbol(&sfn, HIP_SYMBOL(store_cb_ptr), sizeof(void*)));
    void* ldat = dd; void* sdat = ds;
    rocfft_status s1 = rocfft_execution_info_set_load_callback(i1, &lfn, &ldat, 0);
    rocfft_status s2 = rocfft_execution_info_set_store_callback(i3, &sfn, &sdat, 0);
    std::printf("set_load_callback -> %d, set_store_callback -> %d\n", (int)s1, (int)s2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](rocfft_plan p, void* in, void* out, rocfft_execution_info info, const char* name) -> int
    {
        void* ib[1] = {in}; void* ob[1] = {out};
        for (int n=0; n<3; ++n) CHK(rocfft_execute(p, ib, ob, info));
        hipEventRecord(e0);
        for (int n=0; n<10; ++n) CHK(rocfft_execute(p, ib, ob, info));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); std::printf("%-40s %.3f ms\n", name, ms/10);
        return 0;
    };
    // verification first, one execution each on fresh data (real transforms may clobber their input buffer)
    {
        void* ib[1] = {sum}; void* ob[1] = {spec1};
        CHK(rocfft_execute(pf, ib, ob, i0));
        if (s1 == rocfft_status_success) { void* ib2[1] = {a}; void* ob2[1] = {spec2}; CHK(rocfft_execute(pf, ib2, ob2, i1)); }
        CHK(hipDeviceSynchronize());
        std::vector<double2> h1(ncx), h2(ncx);
        CHK(hipMemcpy(h1.data(), spec1, ncx*16, hipMemcpyDeviceToHost)); CHK(hipMemcpy(h2.data(), spec2, ncx*16, hipMemcpyDeviceToHost));
        double dmax = 0, vmax = 0; for (size_t n=0; n<ncx; ++n) { dmax = std::fmax(dmax, std::fmax(std::fabs(h1[n].x-h2[n].x), std::fabs(h1[n].y-h2[n].y))); vmax = std::fmax(vmax, std::fabs(h1[n].x)); }
        std::printf("forward: max |plain - callback| = %.3e (max |spec| %.3e)\n", dmax, vmax);
        // a, b must be intact after the callback run
        std::vector<double> chk(nreal); CHK(hipMemcpy(chk.data(), a, nreal*8, hipMemcpyDeviceToHost));
        double amax = 0; for (size_t n=0; n<nreal; ++n) amax = std::fmax(amax, std::fabs(chk[n] - std::sin(0.001*n)));
        std::printf("input a after the callback transform: max change %.3e\n", amax);
        double2* spec3; CHK(hipMalloc(&spec3, ncx*16)); CHK(hipMemcpy(spec3, spec1, ncx*16, hipMemcpyDeviceToDevice));
        void* ib3[1] = {spec1}; void* ob3[1] = {out1}; CHK(rocfft_execute(pb, ib3, ob3, i2));
        if (s2 == rocfft_status_success) { void* ib4[1] = {spec3}; void* ob4[1] = {out2}; CHK(rocfft_execute(pb, ib4, ob4, i3)); }
        CHK(hipDeviceSynchronize());
        std::vector<double> o1(nreal), o2(nreal);
        CHK(hipMemcpy(o1.data(), out1, nreal*8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(o2.data(), out2, nreal*8, hipMemcpyDeviceToHost));
        dmax = 0; double emax = 0;
        for (size_t n=0; n<nreal; ++n) { dmax = std::fmax(dmax, std::fabs(o1[n]/(nx*ny) - o2[n])); emax = std::fmax(emax, std::fabs(o2[n] - ha[n])); }
        std::printf("inverse: max |plain/N - callback| = %.3e ; round trip error %.3e\n", dmax, emax);
    }
    if (timeit(pf, sum, spec1, i0, "forward, plain")) return 1;
    if (s1 == rocfft_status_success && timeit(pf, a, spec2, i1, "forward, load callback 0.5*(a+b)")) return 1;
    if (timeit(pb, spec1, out1, i2, "inverse, plain")) return 1;
    if (s2 == rocfft_status_success && timeit(pb, spec1, out2, i3, "inverse, store callback (x 1/N)")) return 1;
    return 0;
}
