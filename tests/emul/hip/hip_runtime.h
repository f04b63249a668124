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
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
// dynamic LDS: one buffer of a CU's 160 KB (blocks run one at a time)
extern double g_emul_dyn_lds[160*1024/8];
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(g_emul_dyn_lds);
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return 0; }
struct hipFuncAttributes { size_t localSizeBytes = 0; };
inline hipError_t hipFuncGetAttributes(hipFuncAttributes* a, const void*) { a->localSizeBytes = 0; return 0; }

// Execution model: every simulated thread of a block is a fiber (ucontext). A fiber runs until it finishes or
// reaches __syncthreads(), which yields to the scheduler; the scheduler resumes the block's fibers round-robin,
// so all of them arrive at a barrier before any leaves it. Fibers are started in REVERSE thread order so that
// thread (0,0) goes last within each phase (tests/emul/wave_reduce.h relies on that). Blocks run one at a time,
// so `__shared__` (= static) storage is naturally per block.
#include <ucontext.h>
#include <vector>
#include <functional>
struct EmulFiber { ucontext_t ctx; std::vector<char> stack; bool done = false; uint3e tid; };
struct EmulState { ucontext_t sched; EmulFiber* cur = nullptr; std::function<void()> body; };
extern EmulState g_emul;
extern "C" void emul_fiber_entry();
// The first block of every launch site runs on fibers and records whether any thread reached a barrier; if none
// did, later blocks of that launch site run as plain calls (fast path).
extern bool g_emul_saw_barrier;
inline void __syncthreads() { g_emul_saw_barrier = true; swapcontext(&g_emul.cur->ctx, &g_emul.sched); }

template<class F>
inline void emul_launch(dim3 grid, dim3 block, F&& body)
{
    gridDim = grid; blockDim = block;
    const size_t nthreads = (size_t)block.x*block.y*block.z;
    static std::vector<EmulFiber> fibers;
    if (fibers.size() < nthreads) fibers.resize(nthreads);
    g_emul.body = [&]{ body(); };
    static int need_fibers = -1;              // one flag per launch site (this template is instantiated per kernel lambda): -1 unknown
    for (unsigned bz=0; bz<grid.z; ++bz) for (unsigned by=0; by<grid.y; ++by) for (unsigned bx=0; bx<grid.x; ++bx)
    {
        blockIdx = {bx, by, bz};
        if (need_fibers == 0)
        {
            g_emul.cur = nullptr;
            for (int tz=(int)block.z-1; tz>=0; --tz) for (int ty=(int)block.y-1; ty>=0; --ty) for (int tx=(int)block.x-1; tx>=0; --tx)
            {
                threadIdx = {(unsigned)tx, (unsigned)ty, (unsigned)tz};
                body();
            }
            continue;
        }
        if (need_fibers < 0) g_emul_saw_barrier = false;
        size_t n = 0;
        for (int tz=(int)block.z-1; tz>=0; --tz) for (int ty=(int)block.y-1; ty>=0; --ty) for (int tx=(int)block.x-1; tx>=0; --tx)
        {
            EmulFiber& f = fibers[n++];
            if (f.stack.empty()) f.stack.resize(256*1024);
            f.done = false; f.tid = {(unsigned)tx, (unsigned)ty, (unsigned)tz};
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack.data(); f.ctx.uc_stack.ss_size = f.stack.size(); f.ctx.uc_link = &g_emul.sched;
            makecontext(&f.ctx, (void(*)())emul_fiber_entry, 0);
        }
        bool alive = true;
        while (alive)
        {
            alive = false;
            for (size_t m=0; m<nthreads; ++m)
            {
                EmulFiber& f = fibers[m];
                if (f.done) continue;
                g_emul.cur = &f; threadIdx = f.tid;
                swapcontext(&g_emul.sched, &f.ctx);
                if (!f.done) alive = true;
            }
        }
        if (need_fibers < 0) need_fibers = g_emul_saw_barrier ? 1 : 0;
    }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emul_launch(grid, block, [&]{ kernel(__VA_ARGS__); })
