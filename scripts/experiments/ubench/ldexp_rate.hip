// Issue rate of v_ldexp_f64 against v_mul_f64 / v_add_f64 / v_fma_f64 on gfx950: 8 independent chains per lane, N iterations.
// hipcc --offload-arch=gfx950 -O3 ldexp_rate.hip -o ldexp_rate && ./ldexp_rate
// MI355X: all four 4.2 cycles per wave instruction per SIMD (full rate) -- the v_ldexp_f64 the compiler makes of multiplications by
// powers of two (110 per level in rhs44_march_kernel) cost what the v_mul_f64 would.
#include <hip/hip_runtime.h>
#include <cstdio>
template<int OP> __global__ void k(double* out, double s, int n)
{
    double a[8];
    for (int i=0; i<8; ++i) a[i] = out[threadIdx.x + i*64] + i;
    for (int it=0; it<n; ++it)
    {
#pragma unroll
        for (int i=0; i<8; ++i)
        {
            if (OP == 0) asm volatile("v_ldexp_f64 %0, %1, -4" : "=v"(a[i]) : "v"(a[i]));
            if (OP == 1) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "s"(s));
            if (OP == 2) asm volatile("v_add_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "s"(s));
            if (OP == 3) asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(a[i]) : "v"(a[i]), "s"(s));
        }
    }
    double r = 0; for (int i=0; i<8; ++i) r += a[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = r;
}
int main()
{
    double* d; hipMalloc(&d, 1 << 24); hipMemset(d, 0, 1 << 24);
    const int n = 4000, blocks = 256*8, threads = 256;      // 8 waves per SIMD
    const char* name[4] = {"v_ldexp_f64", "v_mul_f64", "v_add_f64", "v_fma_f64"};
    for (int op=0; op<4; ++op)
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep=0; rep<2; ++rep)
        {
            hipEventRecord(a);
            if (op == 0) k<0><<<blocks, threads>>>(d, 0.0625, n);
            if (op == 1) k<1><<<blocks, threads>>>(d, 0.0625, n);
            if (op == 2) k<2><<<blocks, threads>>>(d, 0.0625, n);
            if (op == 3) k<3><<<blocks, threads>>>(d, 0.0625, n);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        const double inst = (double)blocks*threads/64*8*n;           // wave instructions
        printf("%-12s %.3f ms  %.2f cycles per wave instruction per SIMD at 2.1 GHz\n", name[op], ms, ms*1e-3*2.1e9/(inst/1024));
    }
    return 0;
}
