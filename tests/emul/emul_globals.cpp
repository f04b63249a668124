#include <hip/hip_runtime.h>
thread_local uint3e blockIdx, threadIdx;
thread_local dim3 blockDim, gridDim;
