// tests/emul/hip/hip_runtime.h -- TEST-ONLY stand-in for the HIP runtime so that the library's .hip sources can
// be compiled with g++ and executed on the CPU (tests/emul/libmhh_emul.so). Kernels run one simulated thread
// at a time, block by block; "device" memory is host memory. This exists to catch indexing / logic errors in
// the kernels before a GPU is involved. It is never built into, linked with, or loaded by microhh_amd.
#pragma once
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdint>

struct dim3 { unsigned x, y, z; dim3(unsigned x_=1, unsigned y_=1, unsigned z_=1) : x(x_), y(y_), z(z_) {} };
struct uint3e { unsigned x, y, z; };
extern thread_local uint3e blockIdx, threadIdx;
extern thread_local dim3 blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void* p) { free(p); return 0; }

// Threads of a block run in REVERSE order so that thread (0,0) goes last (see tests/emul/wave_reduce.h).
template<class F>
inline void emul_launch(dim3 grid, dim3 block, F&& body)
{
    gridDim = grid; blockDim = block;
    for (unsigned bz=0; bz<grid.z; ++bz) for (unsigned by=0; by<grid.y; ++by) for (unsigned bx=0; bx<grid.x; ++bx)
    {
        blockIdx = {bx, by, bz};
        for (int tz=(int)block.z-1; tz>=0; --tz) for (int ty=(int)block.y-1; ty>=0; --ty) for (int tx=(int)block.x-1; tx>=0; --tx)
        {
            threadIdx = {(unsigned)tx, (unsigned)ty, (unsigned)tz};
            body();
        }
    }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emul_launch(grid, block, [&]{ kernel(__VA_ARGS__); })
