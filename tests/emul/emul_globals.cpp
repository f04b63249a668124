#include <hip/hip_runtime.h>
thread_local uint3e blockIdx, threadIdx;
thread_local dim3 blockDim, gridDim;
EmulState g_emul;
bool g_emul_saw_barrier = false;
extern "C" void emul_fiber_entry()
{
    g_emul.body();
    g_emul.cur->done = true;     // returning switches to uc_link = the scheduler
}
alignas(64) double g_emul_dyn_lds[160*1024/8];
