// tests/emul/rocfft/rocfft.h -- TEST-ONLY stand-in for the slice of rocFFT that k_pres.hip uses, evaluated with a
// plain O(n^2) DFT on host memory: batched 1-D/2-D real-to-hermitian and hermitian-to-real transforms on
// data with unit element stride, a row stride (2-D) and a batch distance as set by set_data_layout. Lets the pressure solver's own kernels
// (input, column solves, unpack, output) be exercised on the CPU.
#pragma once
#include <cstddef>
#include <cmath>
#include <complex>
#include <vector>
enum rocfft_status { rocfft_status_success = 0, rocfft_status_failure = 1 };
enum rocfft_result_placement { rocfft_placement_inplace, rocfft_placement_notinplace };
enum rocfft_transform_type { rocfft_transform_type_complex_forward, rocfft_transform_type_complex_inverse, rocfft_transform_type_real_forward, rocfft_transform_type_real_inverse };
enum rocfft_precision { rocfft_precision_single, rocfft_precision_double };
enum rocfft_array_type { rocfft_array_type_complex_interleaved, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved };
struct rocfft_plan_description_t { size_t is1 = 0, idist = 0, os1 = 0, odist = 0; };     // row stride / batch distance, 0 = contiguous
typedef rocfft_plan_description_t* rocfft_plan_description;
struct rocfft_plan_t { rocfft_transform_type type; rocfft_precision prec; size_t nd, n0, n1, batch; rocfft_plan_description_t lay; };
typedef rocfft_plan_t* rocfft_plan;
struct rocfft_execution_info_t { void* load_fn = nullptr; void* load_data = nullptr; void* store_fn = nullptr; void* store_data = nullptr; };
typedef rocfft_execution_info_t* rocfft_execution_info;
inline rocfft_status rocfft_setup() { return rocfft_status_success; }
inline rocfft_status rocfft_cleanup() { return rocfft_status_success; }
inline rocfft_status rocfft_plan_description_create(rocfft_plan_description* d) { *d = new rocfft_plan_description_t(); return rocfft_status_success; }
inline rocfft_status rocfft_plan_description_destroy(rocfft_plan_description d) { delete d; return rocfft_status_success; }
inline rocfft_status rocfft_plan_description_set_data_layout(rocfft_plan_description d, rocfft_array_type, rocfft_array_type, const size_t*, const size_t*,
                                                             size_t ins, const size_t* istr, size_t idist, size_t ons, const size_t* ostr, size_t odist)
{
    if ((ins && istr[0] != 1) || (ons && ostr[0] != 1)) return rocfft_status_failure;       // unit element stride only
    d->is1 = ins > 1 ? istr[1] : 0; d->idist = idist; d->os1 = ons > 1 ? ostr[1] : 0; d->odist = odist;
    return rocfft_status_success;
}
inline rocfft_status rocfft_plan_create(rocfft_plan* p, rocfft_result_placement, rocfft_transform_type t, rocfft_precision pr, size_t nd, const size_t* len, size_t batch, rocfft_plan_description d)
{ *p = new rocfft_plan_t{t, pr, nd, len[0], nd > 1 ? len[1] : 1, batch, d ? *d : rocfft_plan_description_t()}; return rocfft_status_success; }
inline rocfft_status rocfft_plan_destroy(rocfft_plan p) { delete p; return rocfft_status_success; }
inline rocfft_status rocfft_execution_info_create(rocfft_execution_info* i) { *i = new rocfft_execution_info_t(); return rocfft_status_success; }
// callbacks (k_pres.hip's fused form): Tdata load(Tdata*, size_t offset, void* cbdata, void*) / void store(Tdata*, size_t, Tdata, void*, void*)
inline rocfft_status rocfft_execution_info_set_load_callback(rocfft_execution_info i, void** fn, void** data, size_t)
{ i->load_fn = fn ? fn[0] : nullptr; i->load_data = data ? data[0] : nullptr; return rocfft_status_success; }
inline rocfft_status rocfft_execution_info_set_store_callback(rocfft_execution_info i, void** fn, void** data, size_t)
{ i->store_fn = fn ? fn[0] : nullptr; i->store_data = data ? data[0] : nullptr; return rocfft_status_success; }
inline rocfft_status rocfft_execution_info_destroy(rocfft_execution_info i) { delete i; return rocfft_status_success; }
inline rocfft_status rocfft_plan_get_work_buffer_size(rocfft_plan, size_t* n) { *n = 0; return rocfft_status_success; }
inline rocfft_status rocfft_execution_info_set_work_buffer(rocfft_execution_info, void*, size_t) { return rocfft_status_success; }
inline rocfft_status rocfft_execution_info_set_stream(rocfft_execution_info, void*) { return rocfft_status_success; }
template<class T>
inline void emul_fft_run(const rocfft_plan_t& P, void* in, void* out, const rocfft_execution_info_t* info)
{
    typedef T (*load_t)(T*, size_t, void*, void*);
    typedef void (*store_t)(T*, size_t, T, void*, void*);
    const load_t lcb = info ? reinterpret_cast<load_t>(info->load_fn) : nullptr;
    const store_t scb = info ? reinterpret_cast<store_t>(info->store_fn) : nullptr;
    typedef std::complex<double> cd;
    const size_t n0 = P.n0, n1 = P.n1, nh = n0/2+1;
    const bool fwd_r = (P.type == rocfft_transform_type_real_forward);
    // hermitian side: row pitch and batch distance (real side and complex transforms: contiguous, checked below)
    const size_t hp = fwd_r ? (P.lay.os1 ? P.lay.os1 : nh) : (P.lay.is1 ? P.lay.is1 : nh);
    const size_t hd = fwd_r ? (P.lay.odist ? P.lay.odist : nh*n1) : (P.lay.idist ? P.lay.idist : nh*n1);
    const double pi = std::acos(-1.0);
    for (size_t b=0; b<P.batch; ++b)
    {
        std::vector<cd> full(n0*n1);
        if (P.type == rocfft_transform_type_complex_forward || P.type == rocfft_transform_type_complex_inverse)
        {
            const double sgn = (P.type == rocfft_transform_type_complex_forward) ? -1.0 : 1.0;
            const std::complex<T>* x = static_cast<const std::complex<T>*>(in) + b*n0;
            std::complex<T>* y = static_cast<std::complex<T>*>(out) + b*n0;
            std::vector<cd> res(n0);
            for (size_t k=0; k<n0; ++k)
            {
                cd acc = 0;
                for (size_t j=0; j<n0; ++j) acc += cd(x[j].real(), x[j].imag()) * std::polar(1.0, sgn*2*pi*(double)((k*j) % n0)/n0);
                res[k] = acc;
            }
            for (size_t k=0; k<n0; ++k) y[k] = std::complex<T>((T)res[k].real(), (T)res[k].imag());
        }
        else if (P.type == rocfft_transform_type_real_forward)
        {
            std::vector<T> rin;
            if (lcb) { rin.resize(n0*n1); for (size_t e=0; e<n0*n1; ++e) rin[e] = lcb(static_cast<T*>(in), b*n0*n1 + e, info->load_data, nullptr); }
            const T* r = lcb ? rin.data() : static_cast<const T*>(in) + b*n0*n1;
            std::complex<T>* h = static_cast<std::complex<T>*>(out) + b*hd;
            for (size_t ky=0; ky<n1; ++ky) for (size_t kx=0; kx<nh; ++kx)
            {
                cd acc = 0;
                for (size_t j=0; j<n1; ++j) for (size_t i=0; i<n0; ++i)
                    acc += (double)r[i + j*n0] * std::polar(1.0, -2*pi*((double)((kx*i) % n0)/n0 + (double)((ky*j) % n1)/n1));
                h[kx + ky*hp] = std::complex<T>((T)acc.real(), (T)acc.imag());
            }
        }
        else
        {
            const std::complex<T>* h = static_cast<const std::complex<T>*>(in) + b*hd;
            T* r = static_cast<T*>(out) + b*n0*n1;
            for (size_t ky=0; ky<n1; ++ky) for (size_t kx=0; kx<n0; ++kx)
            {
                if (kx < nh) full[kx + ky*n0] = cd(h[kx + ky*hp].real(), h[kx + ky*hp].imag());
                else { const size_t mx = n0-kx, my = (n1-ky) % n1; full[kx + ky*n0] = std::conj(cd(h[mx + my*hp].real(), h[mx + my*hp].imag())); }
            }
            for (size_t j=0; j<n1; ++j) for (size_t i=0; i<n0; ++i)
            {
                cd acc = 0;
                for (size_t ky=0; ky<n1; ++ky) for (size_t kx=0; kx<n0; ++kx)
                    acc += full[kx + ky*n0] * std::polar(1.0, 2*pi*((double)((kx*i) % n0)/n0 + (double)((ky*j) % n1)/n1));
                if (scb) scb(static_cast<T*>(out), b*n0*n1 + i + j*n0, (T)acc.real(), info->store_data, nullptr);
                else r[i + j*n0] = (T)acc.real();
            }
        }
    }
}
inline rocfft_status rocfft_execute(rocfft_plan p, void* in[], void* out[], rocfft_execution_info info)
{
    void* o = out ? out[0] : in[0];      // in-place plans pass no output buffer
    if (p->prec == rocfft_precision_double) emul_fft_run<double>(*p, in[0], o, info); else emul_fft_run<float>(*p, in[0], o, info);
    return rocfft_status_success;
}
