// rt_kernel_fast.hip -- fused-arithmetic instance of the path-trace kernel
// (-ffp-contract=fast, hardware rcp/rsq/sqrt/sin/cos/exp2/log2).  Gated by PSNR >= 50 dB
// against the parity instance at equal spp (tests/test_gpu_parity.py).
#define RT_FAST 1
#define RT_NS fast
#define RT_KERNEL_NAME rt_trace_fast
#include "rt_trace.inc.h"

namespace rt {

hipError_t launch_fast(const LaunchParams &p, dim3 grid, size_t lds, hipStream_t stream) {
    hipLaunchKernelGGL(fast::rt_trace_fast, grid, dim3(kBlockThreads), lds, stream, p);
    return hipGetLastError();
}

hipError_t prepare_fast() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(fast::rt_trace_fast),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace rt
