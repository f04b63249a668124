// Which step of evisc_value differs between device and host? Same header (cell_ops.h), same flags as the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "k_common.h"
using namespace mhh;
struct Out { double fac, rit0, rit1, sq1, sq2, ev; };
__host__ __device__ inline Out chain(double s2, double n2, double m0, double z, double z0, double tPr)
{
    Out o;
    o.fac = evisc_mlen2<double>(1, 0, m0, z, z0);
    o.rit0 = n2 / s2;
    o.rit1 = o.rit0 / tPr;
    const double rit = tmin(o.rit1, 1.-1.e-9);
    o.sq1 = dsqrt2(s2); o.sq2 = dsqrt2(1.-rit);
    o.ev = evisc_value<double>(s2, n2, 1, 0, m0, z, z0, tPr);
    return o;
}
__global__ void k(const double* s2, const double* n2, const double* m0, const double* z, double z0, double tPr, Out* o, int n)
{
    const int i = blockIdx.x*256 + threadIdx.x;
    if (i < n) o[i] = chain(s2[i], n2[i], m0[i], z[i], z0, tPr);
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> s2(n), n2(n), m0(n), z(n); std::vector<Out> o(n);
    srand48(3);
    for (int i=0; i<n; ++i) { s2[i] = 1e-6 + drand48()*1e-2; n2[i] = (drand48() - 0.3)*1e-4; m0[i] = 1. + drand48()*20; z[i] = 1. + drand48()*1000; }
    double *a, *b, *c, *d; Out* dO;
    hipMalloc(&a, n*8); hipMalloc(&b, n*8); hipMalloc(&c, n*8); hipMalloc(&d, n*8); hipMalloc(&dO, n*sizeof(Out));
    hipMemcpy(a, s2.data(), n*8, hipMemcpyHostToDevice); hipMemcpy(b, n2.data(), n*8, hipMemcpyHostToDevice);
    hipMemcpy(c, m0.data(), n*8, hipMemcpyHostToDevice); hipMemcpy(d, z.data(), n*8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n/256), dim3(256), 0, 0, a, b, c, d, 0.1, 1./3., dO, n);
    hipMemcpy(o.data(), dO, n*sizeof(Out), hipMemcpyDeviceToHost);
    long w[6] = {0,0,0,0,0,0};
    auto ulp = [](double x, double y) { long long u, v; memcpy(&u, &x, 8); memcpy(&v, &y, 8); return (long)llabs(u - v); };
    for (int i=0; i<n; ++i)
    {
        const Out h = chain(s2[i], n2[i], m0[i], z[i], 0.1, 1./3.);
        const long u[6] = {ulp(o[i].fac, h.fac), ulp(o[i].rit0, h.rit0), ulp(o[i].rit1, h.rit1), ulp(o[i].sq1, h.sq1), ulp(o[i].sq2, h.sq2), ulp(o[i].ev, h.ev)};
        for (int m=0; m<6; ++m) if (u[m] > w[m]) w[m] = u[m];
    }
    printf("worst ulp device vs host: mlen2 %ld, n2/s2 %ld, /tPr %ld, sqrt(s2) %ld, sqrt(1-rit) %ld, evisc %ld\n", w[0], w[1], w[2], w[3], w[4], w[5]);
    return 0;
}
