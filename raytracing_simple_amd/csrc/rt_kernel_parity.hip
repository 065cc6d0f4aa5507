// rt_kernel_parity.hip -- strict-arithmetic instance of the path-trace kernel.
// MUST be compiled with -ffp-contract=off (see _build.py); the pragma below is a second lock.
#pragma clang fp contract(off)
#define RT_FAST 0
#define RT_NS parity
#define RT_KERNEL_NAME rt_trace_parity
#include "rt_trace.inc.h"

namespace rt {

hipError_t launch_parity(const LaunchParams &p, dim3 grid, size_t lds, hipStream_t stream) {
    hipLaunchKernelGGL(parity::rt_trace_parity, grid, dim3(kBlockThreads), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_eval_parity(int op, const float *in, float *out, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(parity::rt_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       stream, op, in, out, n);
    return hipGetLastError();
}

hipError_t prepare_parity() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(parity::rt_trace_parity),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace rt
