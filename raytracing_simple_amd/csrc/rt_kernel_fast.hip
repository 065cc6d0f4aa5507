// rt_kernel_fast.hip -- fused-arithmetic instances of the path-trace kernel
// (-ffp-contract=fast, hardware rcp/rsq/sqrt/sin/cos/exp2/log2).  Gated by PSNR >= 50 dB
// against the parity instance at equal spp (tests/test_gpu_parity.py).  Same set of shipped instances as
// rt_kernel_parity.hip; the A/B shapes exist in the diagnostics build only.
#define RT_FAST 1
#ifndef RT_DIAGNOSTICS
#define RT_DIAGNOSTICS 0
#endif

#define RT_NS fast
#define RT_KERNEL_NAME rt_trace_fast
#define RT_PACK_KERNEL_NAME rt_pack_fast
#if !RT_DIAGNOSTICS
#define RT_NO_RENDER_KERNEL 1        /* as in rt_kernel_parity.hip: the 4-wavefront plain sweep is an A/B shape */
#endif
#include "rt_trace.inc.h"
#if RT_DIAGNOSTICS
#define RT_SCHED_KERNEL_NAME rt_sched_fast
#include "rt_sched.inc.h"
#endif
#include "rt_opts_reset.h"

#define RT_VARIANT_KERNEL 1

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
#define RT_OPT_WALK 1
#define RT_OPT_MINWAVES 5            /* as many waves as workgroups of its LDS tables fit a CU */
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fast_pairs_g           /* tables too large for LDS */
#define RT_KERNEL_NAME rt_trace_fast_pairs_g
#define RT_OPT_WALK 1
#define RT_OPT_GLOBAL_TABLES 1
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fast_pairs_m           /* tables beyond LDS whose PAIRS still fit it: pairs staged, slots where they lie in HBM / L2 */
#define RT_KERNEL_NAME rt_trace_fast_pairs_m
#define RT_OPT_WALK 1
#define RT_OPT_GLOBAL_TABLES 2
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fast_g
#define RT_KERNEL_NAME rt_trace_fast_g
#define RT_OPT_GLOBAL_TABLES 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#if RT_DIAGNOSTICS
// fused mode with every decision in parity arithmetic (RT_OPT_EXACT_DECISIONS, rt_trace.inc.h): the instances BASELINE's
// configurations run on, for tools/fast_gate.py
#define RT_NS fastdx_w1
#define RT_KERNEL_NAME rt_trace_fastdx_w1
#define RT_OPT_WG_WAVES 1
#define RT_OPT_EXACT_DECISIONS 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fastdx_coop_w1
#define RT_KERNEL_NAME rt_trace_fastdx_coop_w1
#define RT_OPT_WG_WAVES 1
#define RT_OPT_COOP 1
#define RT_OPT_EXACT_DECISIONS 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fastdx_coop
#define RT_KERNEL_NAME rt_trace_fastdx_coop
#define RT_OPT_COOP 1
#define RT_OPT_EXACT_DECISIONS 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS fastdx_pairs
#define RT_KERNEL_NAME rt_trace_fastdx_pairs
#define RT_OPT_WALK 1
#define RT_OPT_MINWAVES 5
#define RT_OPT_EXACT_DECISIONS 1
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

static const Instance kFastInstances[] = {
#if RT_DIAGNOSTICS
    { fast::rt_trace_fast, "rt_trace_fast", 4, kTabSweepLds, kRolePlain, 0 },
#endif
    { fast_w1::rt_trace_fast_w1, "rt_trace_fast_w1", 1, kTabSweepLds, kRolePlain, 0 },
    { fast_coop::rt_trace_fast_coop, "rt_trace_fast_coop", 4, kTabSweepLds, kRoleCoop, kInstStaticCoop },
    { fast_coop_w1::rt_trace_fast_coop_w1, "rt_trace_fast_coop_w1", 1, kTabSweepLds, kRoleCoop, kInstStaticCoop },
    { fast_pairs::rt_trace_fast_pairs, "rt_trace_fast_pairs", 4, kTabPairsLds, kRolePairs, 0 },
    { fast_pairs_g::rt_trace_fast_pairs_g, "rt_trace_fast_pairs_g", 4, kTabPairsGlobal, kRolePairsGlobal, 0 },
    { fast_pairs_m::rt_trace_fast_pairs_m, "rt_trace_fast_pairs_m", 4, kTabPairsLdsSlotsGlobal, kRolePairsMixed, 0 },
    { fast_g::rt_trace_fast_g, "rt_trace_fast_g", 4, kTabSweepGlobal, kRoleSweepGlobal, 0 },
#if RT_DIAGNOSTICS
    { fast::rt_sched_fast, "rt_sched_fast", 4, kTabSweepLds, kRoleNone, kInstNoTileCost },
    { fastdx_w1::rt_trace_fastdx_w1, "rt_trace_fastdx_w1", 1, kTabSweepLds, kRoleNone, 0 },
    { fastdx_coop_w1::rt_trace_fastdx_coop_w1, "rt_trace_fastdx_coop_w1", 1, kTabSweepLds, kRoleNone, kInstStaticCoop },
    { fastdx_coop::rt_trace_fastdx_coop, "rt_trace_fastdx_coop", 4, kTabSweepLds, kRoleNone, kInstStaticCoop },
    { fastdx_pairs::rt_trace_fastdx_pairs, "rt_trace_fastdx_pairs", 4, kTabPairsLds, kRoleNone, 0 },
    { fast_persist::rt_trace_fast_persist, "rt_trace_fast_persist", 4, kTabSweepLds, kRolePersist, kInstPersistent | kInstNoTileCost },
    { fast_persist_coop::rt_trace_fast_persist_coop, "rt_trace_fast_persist_coop", 4, kTabSweepLds, kRolePersistCoop,
      kInstPersistent | kInstNoTileCost | kInstStaticCoop },
#endif
};

const Instance *fast_instances(int *count) {
    *count = (int)(sizeof(kFastInstances) / sizeof(kFastInstances[0]));
    return kFastInstances;
}

hipError_t launch_pack_fast(const LaunchParams &p, hipStream_t stream) {
    if (p.local_rows <= 0 || p.w <= 0) return hipSuccess;
    hipLaunchKernelGGL(fast::rt_pack_fast, dim3((unsigned)((p.w + 63) / 64), (unsigned)((p.local_rows + 3) / 4)),
                       dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t prepare_fast() {
    for (const Instance &k : kFastInstances) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k.fn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace rt
