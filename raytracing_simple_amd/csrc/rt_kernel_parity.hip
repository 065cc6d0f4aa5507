// rt_kernel_parity.hip -- strict-arithmetic instances of the path-trace kernel.
// MUST be compiled with -ffp-contract=off (see _build.py); the pragma below is a second lock.
// Shipped (librt_hip.so): rt_trace_parity_w1 (fewer than 12 spheres: single-wavefront workgroups), rt_trace_parity_coop_w1 / _coop
// (12 spheres and more: cooperative any-hit, single- / 4-wavefront workgroups), rt_trace_parity_pairs (many small spheres: the
// hierarchy, rt_walk.inc.h), rt_trace_parity_pairs_m / _pairs_g (its tables beyond LDS), rt_trace_parity_g (the plain sweep over a table
// beyond the sweep's LDS budget, read through the scalar cache: more than about 2 500 records of which fewer than 56 are finite small spheres -- the
// fallback that keeps every input renderable).  The 4-wavefront PLAIN sweep (rt_trace_parity) is not shipped since round 6: a scene of fewer than 12 spheres never
// outgrows the single-wavefront workgroup's LDS budget, so nothing selected it (rt_launch.hip).  Everything else exists only in the
// diagnostics build (librt_hip_diag.so, -DRT_DIAGNOSTICS=1): verification, census and A/B shapes of the same arithmetic
// and the exhaustive device-side checks of the lean square root / reciprocal.  The table at the end of this file is
// the one place that says what each instance is and needs (rt_device.h Instance).
#pragma clang fp contract(off)
#define RT_FAST 0
#ifndef RT_DIAGNOSTICS
#define RT_DIAGNOSTICS 0
#endif

#define RT_NS parity
#define RT_KERNEL_NAME rt_trace_parity
#define RT_PACK_KERNEL_NAME rt_pack_parity
#define RT_OPT_MINWAVES 6            /* <= 80 VGPRs: 6 wavefronts per SIMD */
#if !RT_DIAGNOSTICS
#define RT_NO_RENDER_KERNEL 1        /* product: helpers and the pack kernel only (no scene selects the 4-wavefront plain sweep) */
#endif
#include "rt_trace.inc.h"
#if RT_DIAGNOSTICS
#undef RT_OPT_MINWAVES
#define RT_OPT_MINWAVES 1
#define RT_SCHED_KERNEL_NAME rt_sched_parity
#include "rt_sched.inc.h"
#endif
#include "rt_opts_reset.h"

#define RT_VARIANT_KERNEL 1          /* the pack / scalar-op kernels exist once, above */

#define RT_NS parity_coop
#define RT_KERNEL_NAME rt_trace_parity_coop
#define RT_OPT_COOP 1
#define RT_OPT_MINWAVES 6
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_w1              /* single-wavefront workgroups (8x8 tiles): scenes with small tables */
#define RT_KERNEL_NAME rt_trace_parity_w1
#define RT_OPT_WG_WAVES 1
#define RT_OPT_MINWAVES 6
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_coop_w1
#define RT_KERNEL_NAME rt_trace_parity_coop_w1
#define RT_OPT_WG_WAVES 1
#define RT_OPT_COOP 1
#define RT_OPT_MINWAVES 6
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_pairs           /* large scenes: the walk over sibling pairs, nearer child first (rt_walk.inc.h) */
#define RT_KERNEL_NAME rt_trace_parity_pairs
#define RT_WALK_RAYS_KERNEL_NAME rt_walk_rays_parity   /* diagnostics build: rays through the walk and the sweep */
#define RT_OPT_WALK 1
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_pairs_g         /* tables too large for LDS: the same walk over pairs and slots where they lie in HBM / L2 */
#define RT_KERNEL_NAME rt_trace_parity_pairs_g
#define RT_OPT_WALK 1
#define RT_OPT_GLOBAL_TABLES 1
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_pairs_m          /* tables beyond LDS whose PAIRS still fit it: pairs staged, slots where they lie in HBM / L2 */
#define RT_KERNEL_NAME rt_trace_parity_pairs_m
#define RT_OPT_WALK 1
#define RT_OPT_GLOBAL_TABLES 2
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_g               /* ... and the plain sweep over the table in HBM / L2 (no hierarchy, or it lost the measurement) */
#define RT_KERNEL_NAME rt_trace_parity_g
#define RT_OPT_GLOBAL_TABLES 1
#define RT_OPT_MINWAVES 6
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#if RT_DIAGNOSTICS
#define RT_NS parity_pairs_census    /* the walk with a census of what it executes (counters[20..29], [8..15]) */
#define RT_KERNEL_NAME rt_trace_parity_pairs_census
#define RT_OPT_WALK 2
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_pairs_planes    /* A/B: the staged pairs as four planes of 16-byte parts (LDS bank conflicts of the pair fetch) */
#define RT_KERNEL_NAME rt_trace_parity_pairs_planes
#define RT_OPT_WALK 1
#define RT_OPT_PAIR_PLANES 1
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_pairs_2r        /* A/B: the hierarchy walk with TWO pixels per lane (RT_OPT_RAYS2): both rays' state in registers, 3 wavefronts per SIMD */
#define RT_KERNEL_NAME rt_trace_parity_pairs_2r
#define RT_OPT_WALK 1
#define RT_OPT_RAYS2 1
#define RT_OPT_MINWAVES 3
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_pairs_2r_census /* ... and its census (steps, lanes per step) */
#define RT_KERNEL_NAME rt_trace_parity_pairs_2r_census
#define RT_OPT_WALK 2
#define RT_OPT_RAYS2 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_pairs_g_w4      /* A/B: the L2 walk under the 4-waves launch bound it shipped with until round 5 (98 registers then) */
#define RT_KERNEL_NAME rt_trace_parity_pairs_g_w4
#define RT_OPT_WALK 1
#define RT_OPT_GLOBAL_TABLES 1
#define RT_OPT_MINWAVES 4
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_pairs_gt        /* A/B: the L2 walk with the promoted top of the tree (BvhTables::n_top pairs) staged in LDS */
#define RT_KERNEL_NAME rt_trace_parity_pairs_gt
#define RT_OPT_WALK 1
#define RT_OPT_GLOBAL_TABLES 1
#define RT_OPT_TOP_PAIRS 1
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_pairs_gp        /* A/B: the L2 walk with both children's records requested one level ahead */
#define RT_KERNEL_NAME rt_trace_parity_pairs_gp
#define RT_OPT_WALK 1
#define RT_OPT_GLOBAL_TABLES 1
#define RT_OPT_PREFETCH 1
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_pairs_gtp       /* ... both arms together */
#define RT_KERNEL_NAME rt_trace_parity_pairs_gtp
#define RT_OPT_WALK 1
#define RT_OPT_GLOBAL_TABLES 1
#define RT_OPT_TOP_PAIRS 1
#define RT_OPT_PREFETCH 1
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_pairs_gq        /* A/B: the L2 walk over the PACKED pair table (32 bytes per pair step instead of 64) */
#define RT_KERNEL_NAME rt_trace_parity_pairs_gq
#define RT_OPT_WALK 1
#define RT_OPT_GLOBAL_TABLES 1
#define RT_OPT_PACKED_PAIRS 1
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_coop_check      /* coop + the sequential sweep beside it (verification) */
#define RT_KERNEL_NAME rt_trace_parity_coop_check
#define RT_OPT_COOP 2
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_census          /* section census (tools/stamp_profile.py) */
#define RT_KERNEL_NAME rt_trace_parity_census
#define RT_SCHED_KERNEL_NAME rt_sched_parity_census
#define RT_OPT_STAMPS 1
#include "rt_trace.inc.h"
#include "rt_sched.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_coop_census     /* ... of the cooperative any-hit instance (12 spheres and more: the 16-sphere scene, C5) */
#define RT_KERNEL_NAME rt_trace_parity_coop_census
#define RT_OPT_COOP 1
#define RT_OPT_STAMPS 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_persist
#define RT_KERNEL_NAME rt_trace_parity_persist
#define RT_OPT_PERSIST 1
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_persist_coop
#define RT_KERNEL_NAME rt_trace_parity_persist_coop
#define RT_OPT_PERSIST 1
#define RT_OPT_COOP 1
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_tl              /* the shipped shape + device wall-clock logging (P.timelog / P.wavelog) */
#define RT_KERNEL_NAME rt_trace_parity_tl
#define RT_OPT_TIMELOG 1
#define RT_OPT_MINWAVES 6
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"
#endif   // RT_DIAGNOSTICS

namespace rt {

// what each instance is and needs: { kernel, symbol, wavefronts per workgroup, tables, role, flags }
static const Instance kParityInstances[] = {
#if RT_DIAGNOSTICS
    { parity::rt_trace_parity, "rt_trace_parity", 4, kTabSweepLds, kRolePlain, 0 },      // A/B only (rt_debug_set_wg_waves(ctx, 4) on a scene below 12 spheres)
#endif
    { parity_w1::rt_trace_parity_w1, "rt_trace_parity_w1", 1, kTabSweepLds, kRolePlain, 0 },
    { parity_coop::rt_trace_parity_coop, "rt_trace_parity_coop", 4, kTabSweepLds, kRoleCoop, kInstStaticCoop },
    { parity_coop_w1::rt_trace_parity_coop_w1, "rt_trace_parity_coop_w1", 1, kTabSweepLds, kRoleCoop, kInstStaticCoop },
    { parity_pairs::rt_trace_parity_pairs, "rt_trace_parity_pairs", 4, kTabPairsLds, kRolePairs, 0 },
    { parity_pairs_g::rt_trace_parity_pairs_g, "rt_trace_parity_pairs_g", 4, kTabPairsGlobal, kRolePairsGlobal, 0 },
    { parity_pairs_m::rt_trace_parity_pairs_m, "rt_trace_parity_pairs_m", 4, kTabPairsLdsSlotsGlobal, kRolePairsMixed, 0 },
    { parity_g::rt_trace_parity_g, "rt_trace_parity_g", 4, kTabSweepGlobal, kRoleSweepGlobal, 0 },
#if RT_DIAGNOSTICS
    { parity_pairs_census::rt_trace_parity_pairs_census, "rt_trace_parity_pairs_census", 4, kTabPairsLds, kRoleNone, 0 },
    { parity_pairs_planes::rt_trace_parity_pairs_planes, "rt_trace_parity_pairs_planes", 4, kTabPairsLds, kRoleNone, 0 },
    { parity_pairs_2r::rt_trace_parity_pairs_2r, "rt_trace_parity_pairs_2r", 4, kTabPairsLds, kRoleNone, kInstTwoRays },
    { parity_pairs_2r_census::rt_trace_parity_pairs_2r_census, "rt_trace_parity_pairs_2r_census", 4, kTabPairsLds, kRoleNone, kInstTwoRays },
    { parity_pairs_g_w4::rt_trace_parity_pairs_g_w4, "rt_trace_parity_pairs_g_w4", 4, kTabPairsGlobal, kRoleNone, 0 },
    { parity_pairs_gt::rt_trace_parity_pairs_gt, "rt_trace_parity_pairs_gt", 4, kTabPairsTopLds, kRoleNone, 0 },
    { parity_pairs_gp::rt_trace_parity_pairs_gp, "rt_trace_parity_pairs_gp", 4, kTabPairsGlobal, kRoleNone, 0 },
    { parity_pairs_gq::rt_trace_parity_pairs_gq, "rt_trace_parity_pairs_gq", 4, kTabPairsPacked, kRoleNone, 0 },
    { parity_pairs_gtp::rt_trace_parity_pairs_gtp, "rt_trace_parity_pairs_gtp", 4, kTabPairsTopLds, kRoleNone, 0 },
    { parity_coop_check::rt_trace_parity_coop_check, "rt_trace_parity_coop_check", 4, kTabSweepLds, kRoleNone, kInstStaticCoop },
    { parity_census::rt_trace_parity_census, "rt_trace_parity_census", 4, kTabSweepLds, kRoleNone, 0 },
    { parity_coop_census::rt_trace_parity_coop_census, "rt_trace_parity_coop_census", 4, kTabSweepLds, kRoleNone, kInstStaticCoop },
    { parity::rt_sched_parity, "rt_sched_parity", 4, kTabSweepLds, kRoleNone, kInstNoTileCost },              // stage-scheduled (in-register ray queue) A/B
    { parity_census::rt_sched_parity_census, "rt_sched_parity_census", 4, kTabSweepLds, kRoleNone, kInstNoTileCost },
    { parity_persist::rt_trace_parity_persist, "rt_trace_parity_persist", 4, kTabSweepLds, kRolePersist, kInstPersistent | kInstNoTileCost },
    { parity_persist_coop::rt_trace_parity_persist_coop, "rt_trace_parity_persist_coop", 4, kTabSweepLds, kRolePersistCoop,
      kInstPersistent | kInstNoTileCost | kInstStaticCoop },
    { parity_tl::rt_trace_parity_tl, "rt_trace_parity_tl", 4, kTabSweepLds, kRoleTimelog, 0 },
#endif
};

const Instance *parity_instances(int *count) {
    *count = (int)(sizeof(kParityInstances) / sizeof(kParityInstances[0]));
    return kParityInstances;
}

hipError_t launch_instance(const Instance &inst, const LaunchParams &p, dim3 grid, size_t lds, hipStream_t stream) {
    if (!inst.fn) return hipErrorInvalidValue;
    hipLaunchKernelGGL(inst.fn, grid, dim3(64 * inst.waves), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_pack_parity(const LaunchParams &p, hipStream_t stream) {
    if (p.local_rows <= 0 || p.w <= 0) return hipSuccess;
    hipLaunchKernelGGL(parity::rt_pack_parity, dim3((unsigned)((p.w + 63) / 64), (unsigned)((p.local_rows + 3) / 4)),
                       dim3(256), 0, stream, p);
    return hipGetLastError();
}

#if RT_DIAGNOSTICS
hipError_t launch_walk_rays(const LaunchParams &p, const float4 *rays, uint32_t n_rays, uint4 *out, size_t lds, hipStream_t stream) {
    if (n_rays == 0) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(parity_pairs::rt_walk_rays_parity),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    if (e != hipSuccess) return e;
    const unsigned blocks = (n_rays + 255) / 256 < 512 ? (n_rays + 255) / 256 : 512;
    hipLaunchKernelGGL(parity_pairs::rt_walk_rays_parity, dim3(blocks), dim3(256), lds, stream, p, rays, n_rays, out);
    return hipGetLastError();
}

hipError_t launch_eval_parity(int op, const float *in, float *out, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(parity::rt_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       stream, op, in, out, n);
    return hipGetLastError();
}

hipError_t launch_sqrt_check(unsigned long long *d_mismatches, hipStream_t stream, int which) {
    if (which == 1) {
        hipLaunchKernelGGL(parity::rt_hitpost_check_kernel, dim3(256 * 16), dim3(256), 0, stream, d_mismatches);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(parity::rt_sqrt_check_kernel, dim3(256 * 16), dim3(256), 0, stream, d_mismatches);
    return hipGetLastError();
}

hipError_t launch_rcp_probe(unsigned long long *d_hist, hipStream_t stream) {
    hipLaunchKernelGGL(parity::rt_rcp_probe_kernel, dim3(256 * 16), dim3(256), 0, stream, d_hist);
    return hipGetLastError();
}
#endif

hipError_t prepare_parity() {
    for (const Instance &k : kParityInstances) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k.fn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace rt
