// rt_kernel_parity.hip -- strict-arithmetic instances of the path-trace kernel.
// MUST be compiled with -ffp-contract=off (see _build.py); the pragma below is a second lock.
// Instance 0 is the shipped one (RT_MODE_PARITY); the others are A/B shapes of the same
// arithmetic, selectable as mode 100+k for in-process comparisons (tools/ab_bench.py).
#pragma clang fp contract(off)
#define RT_FAST 0

#define RT_NS parity
#define RT_KERNEL_NAME rt_trace_parity
#define RT_SCHED_KERNEL_NAME rt_sched_parity
#include "rt_trace.inc.h"
#include "rt_sched.inc.h"
#undef RT_NS
#undef RT_KERNEL_NAME
#undef RT_SCHED_KERNEL_NAME

#define RT_VARIANT_KERNEL 1
#undef RT_OPT_UNROLL
#undef RT_OPT_SKIPNEG

#define RT_NS parity_coop
#define RT_KERNEL_NAME rt_trace_parity_coop
#define RT_OPT_COOP 1
#include "rt_trace.inc.h"
#undef RT_NS
#undef RT_KERNEL_NAME
#undef RT_OPT_COOP

#define RT_NS parity_coopv
#define RT_KERNEL_NAME rt_trace_parity_coopv
#define RT_OPT_COOP 2
#include "rt_trace.inc.h"
#undef RT_NS
#undef RT_KERNEL_NAME
#undef RT_OPT_COOP

#define RT_NS parity_a5
#define RT_KERNEL_NAME rt_trace_parity_a5
#define RT_SCHED_KERNEL_NAME rt_sched_parity_a5
#define RT_OPT_STAMPS 1
#include "rt_trace.inc.h"
#include "rt_sched.inc.h"
#undef RT_NS
#undef RT_KERNEL_NAME
#undef RT_SCHED_KERNEL_NAME
#undef RT_OPT_STAMPS

namespace rt {

using KernelFn = void (*)(const LaunchParams);
static KernelFn const kParityKernels[] = {
    parity::rt_trace_parity, parity_a5::rt_trace_parity_a5,
    parity::rt_sched_parity, parity_a5::rt_sched_parity_a5, parity_coop::rt_trace_parity_coop,
    parity_coopv::rt_trace_parity_coopv,

};
constexpr int kParityCount = sizeof(kParityKernels) / sizeof(kParityKernels[0]);

int parity_variant_count() { return kParityCount; }

hipError_t launch_parity(int variant, const LaunchParams &p, dim3 grid, size_t lds, hipStream_t stream) {
    if (variant < 0 || variant >= kParityCount) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kParityKernels[variant], grid, dim3(kBlockThreads), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_eval_parity(int op, const float *in, float *out, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(parity::rt_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       stream, op, in, out, n);
    return hipGetLastError();
}

hipError_t prepare_parity() {
    for (KernelFn k : kParityKernels) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace rt
