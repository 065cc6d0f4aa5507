// rt_kernel_fast.hip -- fused-arithmetic instances of the path-trace kernel
// (-ffp-contract=fast, hardware rcp/rsq/sqrt/sin/cos/exp2/log2).  Gated by PSNR >= 50 dB
// against the parity instance at equal spp (tests/test_gpu_parity.py).
//   [0] rt_trace_fast, [3] rt_trace_fast_coop, [6] rt_trace_fast_w1, [7] rt_trace_fast_coop_w1, [10] rt_trace_fast_pairs: shipped; the
//   others are A/B shapes (mode 200+k) of the diagnostics build (librt_hip_diag.so, -DRT_DIAGNOSTICS=1).
#define RT_FAST 1
#ifndef RT_DIAGNOSTICS
#define RT_DIAGNOSTICS 0
#endif

#define RT_NS fast
#define RT_KERNEL_NAME rt_trace_fast
#define RT_PACK_KERNEL_NAME rt_pack_fast
#include "rt_trace.inc.h"
#if RT_DIAGNOSTICS
#define RT_SCHED_KERNEL_NAME rt_sched_fast
#include "rt_sched.inc.h"
#endif
#include "rt_opts_reset.h"

#define RT_VARIANT_KERNEL 1

#if RT_DIAGNOSTICS
#define RT_NS fast_a1                /* the round-1 sweep shape: no unroll, no ballot skip */
#define RT_KERNEL_NAME rt_trace_fast_a1
#define RT_OPT_UNROLL 1
#define RT_OPT_SKIPNEG 0
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"
#endif

#define RT_NS fast_coop
#define RT_KERNEL_NAME rt_trace_fast_coop
#define RT_OPT_COOP 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fast_w1
#define RT_KERNEL_NAME rt_trace_fast_w1
#define RT_OPT_WG_WAVES 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fast_coop_w1
#define RT_KERNEL_NAME rt_trace_fast_coop_w1
#define RT_OPT_WG_WAVES 1
#define RT_OPT_COOP 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fast_pairs
#define RT_KERNEL_NAME rt_trace_fast_pairs
#define RT_OPT_BVH 6
#define RT_OPT_MINWAVES 5            /* as many waves as workgroups of its LDS tables fit a CU */
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fast_pairs_g           /* tables too large for LDS */
#define RT_KERNEL_NAME rt_trace_fast_pairs_g
#define RT_OPT_BVH 6
#define RT_OPT_GLOBAL_TABLES 1
#define RT_OPT_MINWAVES 4
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fast_g
#define RT_KERNEL_NAME rt_trace_fast_g
#define RT_OPT_GLOBAL_TABLES 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#if RT_DIAGNOSTICS
#define RT_NS fast_bvh
#define RT_KERNEL_NAME rt_trace_fast_bvh
#define RT_OPT_BVH 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fast_walk
#define RT_KERNEL_NAME rt_trace_fast_walk
#define RT_OPT_BVH 4
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
#endif

namespace rt {

using KernelFn = void (*)(const LaunchParams);
static KernelFn const kFastKernels[] = {
    fast::rt_trace_fast,                 // 0
#if RT_DIAGNOSTICS
    fast_a1::rt_trace_fast_a1,           // 1
    fast::rt_sched_fast,                 // 2
#else
    nullptr, nullptr,
#endif
    fast_coop::rt_trace_fast_coop,       // 3 = kFastCoopVariant
#if RT_DIAGNOSTICS
    fast_persist::rt_trace_fast_persist,             // 4 = kFastPersistVariant
    fast_persist_coop::rt_trace_fast_persist_coop,   // 5 = kFastPersistCoopVariant
#else
    nullptr, nullptr,
#endif
    fast_w1::rt_trace_fast_w1,                       // 6 = kFastW1Variant
    fast_coop_w1::rt_trace_fast_coop_w1,             // 7 = kFastCoopW1Variant
#if RT_DIAGNOSTICS
    fast_bvh::rt_trace_fast_bvh,                     // 8 = kFastBvhVariant     A/B forms of the hierarchy walk
    fast_walk::rt_trace_fast_walk,                   // 9 = kFastWalkVariant
#else
    nullptr, nullptr,
#endif
    fast_pairs::rt_trace_fast_pairs,                 // 10 = kFastPairsVariant  shipped: large scenes
    fast_pairs_g::rt_trace_fast_pairs_g,             // 11 = kFastPairsGlobalVariant   shipped: tables beyond LDS
    fast_g::rt_trace_fast_g,                         // 12 = kFastGlobalVariant
};
constexpr int kFastCount = sizeof(kFastKernels) / sizeof(kFastKernels[0]);

int fast_variant_count() { return kFastCount; }
const char *fast_variant_name(int variant) {
    switch (variant) {
        case 0: return "rt_trace_fast";
        case kFastCoopVariant: return "rt_trace_fast_coop";
        case kFastW1Variant: return "rt_trace_fast_w1";
        case kFastCoopW1Variant: return "rt_trace_fast_coop_w1";
        case kFastPairsVariant: return "rt_trace_fast_pairs";
        case kFastPairsGlobalVariant: return "rt_trace_fast_pairs_g";
        case kFastGlobalVariant: return "rt_trace_fast_g";
        default: return "rt_trace_fast (a diagnostics instance)";
    }
}
int fast_variant_waves(int variant) { return (variant == kFastW1Variant || variant == kFastCoopW1Variant) ? 1 : 4; }

hipError_t launch_fast(int variant, const LaunchParams &p, dim3 grid, size_t lds, hipStream_t stream) {
    if (variant < 0 || variant >= kFastCount || !kFastKernels[variant]) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kFastKernels[variant], grid, dim3(64 * fast_variant_waves(variant)), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_pack_fast(const LaunchParams &p, hipStream_t stream) {
    if (p.local_rows <= 0 || p.w <= 0) return hipSuccess;
    hipLaunchKernelGGL(fast::rt_pack_fast, dim3((unsigned)((p.w + 63) / 64), (unsigned)((p.local_rows + 3) / 4)),
                       dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t prepare_fast() {
    for (KernelFn k : kFastKernels) {
        if (!k) continue;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace rt
