// rt_kernel_fast.hip -- fused-arithmetic instances of the path-trace kernel
// (-ffp-contract=fast, hardware rcp/rsq/sqrt/sin/cos/exp2/log2).  Gated by PSNR >= 50 dB
// against the parity instance at equal spp (tests/test_gpu_parity.py).
//   [0] rt_trace_fast, [3] rt_trace_fast_coop: shipped; the others are A/B shapes (mode 200+k).
#define RT_FAST 1

#define RT_NS fast
#define RT_KERNEL_NAME rt_trace_fast
#include "rt_trace.inc.h"
#define RT_SCHED_KERNEL_NAME rt_sched_fast
#include "rt_sched.inc.h"
#include "rt_opts_reset.h"

#define RT_VARIANT_KERNEL 1

#define RT_NS fast_a1                /* the round-1 sweep shape: no unroll, no ballot skip */
#define RT_KERNEL_NAME rt_trace_fast_a1
#define RT_OPT_UNROLL 1
#define RT_OPT_SKIPNEG 0
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fast_coop
#define RT_KERNEL_NAME rt_trace_fast_coop
#define RT_OPT_COOP 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fast_persist
#define RT_KERNEL_NAME rt_trace_fast_persist
#define RT_OPT_PERSIST 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fast_persist_coop
#define RT_KERNEL_NAME rt_trace_fast_persist_coop
#define RT_OPT_PERSIST 1
#define RT_OPT_COOP 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

namespace rt {

using KernelFn = void (*)(const LaunchParams);
static KernelFn const kFastKernels[] = { fast::rt_trace_fast, fast_a1::rt_trace_fast_a1, fast::rt_sched_fast,
                                           fast_coop::rt_trace_fast_coop, fast_persist::rt_trace_fast_persist,
                                           fast_persist_coop::rt_trace_fast_persist_coop };
constexpr int kFastCount = sizeof(kFastKernels) / sizeof(kFastKernels[0]);

int fast_variant_count() { return kFastCount; }

hipError_t launch_fast(int variant, const LaunchParams &p, dim3 grid, size_t lds, hipStream_t stream) {
    if (variant < 0 || variant >= kFastCount) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kFastKernels[variant], grid, dim3(kBlockThreads), lds, stream, p);
    return hipGetLastError();
}

hipError_t prepare_fast() {
    for (KernelFn k : kFastKernels) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace rt
