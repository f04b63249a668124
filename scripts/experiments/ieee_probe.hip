// Are fp64 sqrt and division of the device (as compiled for the library: -O3 -ffp-contract=off, no fast-math) correctly rounded?
// Compares 4M random operands against the host's results bit for bit.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
__global__ void k(const double* a, const double* b, double* s, double* q, double* r, int n)
{
    const int i = blockIdx.x*256 + threadIdx.x;
    if (i < n) { s[i] = __builtin_sqrt(a[i]); q[i] = a[i] / b[i]; r[i] = 1.0 / b[i]; }
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> a(n), b(n), s(n), q(n), r(n);
    srand48(7);
    for (int i=0; i<n; ++i) { a[i] = std::ldexp(drand48() + 0.5, (int)(lrand48() % 60) - 30); b[i] = std::ldexp(drand48() + 0.5, (int)(lrand48() % 60) - 30); }
    double *da, *db, *ds, *dq, *dr;
    hipMalloc(&da, n*8); hipMalloc(&db, n*8); hipMalloc(&ds, n*8); hipMalloc(&dq, n*8); hipMalloc(&dr, n*8);
    hipMemcpy(da, a.data(), n*8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n*8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n/256), dim3(256), 0, 0, da, db, ds, dq, dr, n);
    hipMemcpy(s.data(), ds, n*8, hipMemcpyDeviceToHost); hipMemcpy(q.data(), dq, n*8, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dr, n*8, hipMemcpyDeviceToHost);
    long bs = 0, bq = 0, br = 0; long ws = 0, wq = 0, wr = 0;
    auto ulp = [](double x, double y) { long long u, v; memcpy(&u, &x, 8); memcpy(&v, &y, 8); return (long)llabs(u - v); };
    for (int i=0; i<n; ++i)
    {
        const long us = ulp(s[i], std::sqrt(a[i])), uq = ulp(q[i], a[i]/b[i]), ur = ulp(r[i], 1.0/b[i]);
        bs += us != 0; bq += uq != 0; br += ur != 0;
        if (us > ws) ws = us; if (uq > wq) wq = uq; if (ur > wr) wr = ur;
    }
    printf("sqrt: %ld of %d differ (worst %ld ulp); a/b: %ld differ (worst %ld ulp); 1/b: %ld differ (worst %ld ulp)\n", bs, n, ws, bq, wq, br, wr);
    return 0;
}
