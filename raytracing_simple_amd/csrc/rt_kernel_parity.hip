// rt_kernel_parity.hip -- strict-arithmetic instances of the path-trace kernel.
// MUST be compiled with -ffp-contract=off (see _build.py); the pragma below is a second lock.
//   [0] rt_trace_parity, [10] rt_trace_parity_w1            shipped: small scenes (4-wave / single-wave workgroups)
//   [4] rt_trace_parity_coop, [11] rt_trace_parity_coop_w1  shipped: scenes with >= 12 spheres (cooperative any-hit)
//   [17] rt_trace_parity_pairs                              shipped: scenes with many small spheres (hierarchy, rt_walk.inc.h)
//   (rt_api.hip launch() takes the single-wavefront shape while the scene tables leave LDS room for 6 waves per SIMD)
// Everything else exists only in the diagnostics build (librt_hip_diag.so, -DRT_DIAGNOSTICS=1):
// A/B and verification shapes of the same arithmetic (mode 100+k, tools/ab_bench.py) and the
// exhaustive device-side checks of the lean square root / reciprocal.
#pragma clang fp contract(off)
#define RT_FAST 0
#ifndef RT_DIAGNOSTICS
#define RT_DIAGNOSTICS 0
#endif

#define RT_NS parity
#define RT_KERNEL_NAME rt_trace_parity
#define RT_PACK_KERNEL_NAME rt_pack_parity
#define RT_OPT_LEAN_SQRT 1
#define RT_OPT_MINWAVES 6            /* <= 80 VGPRs: 6 wavefronts per SIMD */
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
#define RT_OPT_LEAN_SQRT 1
#define RT_OPT_MINWAVES 6
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_w1              /* single-wavefront workgroups (8x8 tiles): scenes with small tables */
#define RT_KERNEL_NAME rt_trace_parity_w1
#define RT_OPT_WG_WAVES 1
#define RT_OPT_LEAN_SQRT 1
#define RT_OPT_MINWAVES 6
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_coop_w1
#define RT_KERNEL_NAME rt_trace_parity_coop_w1
#define RT_OPT_WG_WAVES 1
#define RT_OPT_COOP 1
#define RT_OPT_LEAN_SQRT 1
#define RT_OPT_MINWAVES 6
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_pairs           /* ... over sibling pairs, nearer child first, the other on a per-lane stack */
#define RT_KERNEL_NAME rt_trace_parity_pairs
#define RT_WALK_RAYS_KERNEL_NAME rt_walk_rays_parity   /* diagnostics build: rays through the walk and the sweep */
#define RT_OPT_BVH 6
#define RT_OPT_LEAN_SQRT 1
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_pairs_g         /* tables too large for LDS: the same walk over pairs and slots where they lie in HBM / L2 */
#define RT_KERNEL_NAME rt_trace_parity_pairs_g
#define RT_OPT_BVH 6
#define RT_OPT_GLOBAL_TABLES 1
#define RT_OPT_LEAN_SQRT 1
#define RT_OPT_MINWAVES 4
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_g               /* ... and the plain sweep over the table in HBM / L2 (no hierarchy, or it lost the measurement) */
#define RT_KERNEL_NAME rt_trace_parity_g
#define RT_OPT_GLOBAL_TABLES 1
#define RT_OPT_LEAN_SQRT 1
#define RT_OPT_MINWAVES 6
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#if RT_DIAGNOSTICS
#define RT_NS parity_bvh             /* A/B: depth-first nodes with skip links, walked to the end inside each closest-hit / shadow call */
#define RT_KERNEL_NAME rt_trace_parity_bvh
#define RT_OPT_BVH 1
#define RT_OPT_LEAN_SQRT 1
#define RT_OPT_MINWAVES 4
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_walk            /* A/B: the same nodes, the walk as lane state (rt_walk.inc.h) */
#define RT_KERNEL_NAME rt_trace_parity_walk
#define RT_OPT_BVH 4
#define RT_OPT_LEAN_SQRT 1
#define RT_OPT_MINWAVES 6
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_bvhv            /* the walk + the plain sweep beside it (verification) */
#define RT_KERNEL_NAME rt_trace_parity_bvhv
#define RT_OPT_BVH 2
#define RT_OPT_LEAN_SQRT 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_bvhs            /* the walk with a census of its steps (counters[20..27]) */
#define RT_KERNEL_NAME rt_trace_parity_bvhs
#define RT_OPT_BVH 3
#define RT_OPT_LEAN_SQRT 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_walks           /* rt_walk.inc.h with a census of its two phases (counters[20..28]) */
#define RT_KERNEL_NAME rt_trace_parity_walks
#define RT_OPT_BVH 5
#define RT_OPT_LEAN_SQRT 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_pairss          /* the pair walk with a census of its two phases (counters[20..28]) */
#define RT_KERNEL_NAME rt_trace_parity_pairss
#define RT_OPT_BVH 7
#define RT_OPT_LEAN_SQRT 1
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_coopv           /* coop + the sequential sweep beside it (verification) */
#define RT_KERNEL_NAME rt_trace_parity_coopv
#define RT_OPT_COOP 2
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_a5              /* section census (tools/stamp_profile.py) */
#define RT_KERNEL_NAME rt_trace_parity_a5
#define RT_SCHED_KERNEL_NAME rt_sched_parity_a5
#define RT_OPT_STAMPS 1
#include "rt_trace.inc.h"
#include "rt_sched.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_persist
#define RT_KERNEL_NAME rt_trace_parity_persist
#define RT_OPT_PERSIST 1
#define RT_OPT_LEAN_SQRT 1
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_persist_coop
#define RT_KERNEL_NAME rt_trace_parity_persist_coop
#define RT_OPT_PERSIST 1
#define RT_OPT_COOP 1
#define RT_OPT_LEAN_SQRT 1
#define RT_OPT_MINWAVES 5
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_r0              /* A/B: the shipped shape without its newest change */
#define RT_KERNEL_NAME rt_trace_parity_r0
#define RT_OPT_LEAN_SQRT 1
#define RT_OPT_AB_OLD 1
#define RT_OPT_MINWAVES 6
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"

#define RT_NS parity_tl              /* the shipped shape + device wall-clock logging (P.timelog / P.wavelog) */
#define RT_KERNEL_NAME rt_trace_parity_tl
#define RT_OPT_LEAN_SQRT 1
#define RT_OPT_TIMELOG 1
#define RT_OPT_MINWAVES 6
#include "rt_trace.inc.h"
#include "rt_opts_reset.h"
#endif   // RT_DIAGNOSTICS

namespace rt {

using KernelFn = void (*)(const LaunchParams);
static KernelFn const kParityKernels[] = {
    parity::rt_trace_parity,            // 0
#if RT_DIAGNOSTICS
    parity_a5::rt_trace_parity_a5,      // 1  census
    parity::rt_sched_parity,            // 2  stage-scheduled (in-register queue) A/B
    parity_a5::rt_sched_parity_a5,      // 3  its census
#else
    nullptr, nullptr, nullptr,
#endif
    parity_coop::rt_trace_parity_coop,  // 4  = kParityCoopVariant
#if RT_DIAGNOSTICS
    parity_coopv::rt_trace_parity_coopv,  // 5
    parity_persist::rt_trace_parity_persist,            // 6 = kParityPersistVariant
    parity_persist_coop::rt_trace_parity_persist_coop,  // 7 = kParityPersistCoopVariant
    parity_r0::rt_trace_parity_r0,                      // 8
    parity_tl::rt_trace_parity_tl,                      // 9 = kParityTimelogVariant
#else
    nullptr, nullptr, nullptr, nullptr, nullptr,
#endif
    parity_w1::rt_trace_parity_w1,                      // 10 = kParityW1Variant
    parity_coop_w1::rt_trace_parity_coop_w1,            // 11 = kParityCoopW1Variant
#if RT_DIAGNOSTICS
    parity_bvh::rt_trace_parity_bvh,                    // 12 = kParityBvhVariant     A/B: walk per call
    parity_bvhv::rt_trace_parity_bvhv,                  // 13 = kParityBvhCheckVariant    ... with the plain sweep beside it
    parity_bvhs::rt_trace_parity_bvhs,                  // 14   ... with a census of its steps
    parity_walk::rt_trace_parity_walk,                  // 15 = kParityWalkVariant    A/B: walk as lane state, depth-first nodes
    parity_walks::rt_trace_parity_walks,                // 16   its census
#else
    nullptr, nullptr, nullptr, nullptr, nullptr,
#endif
    parity_pairs::rt_trace_parity_pairs,                // 17 = kParityPairsVariant   shipped: large scenes
#if RT_DIAGNOSTICS
    parity_pairss::rt_trace_parity_pairss,              // 18   its census
#else
    nullptr,
#endif
    parity_pairs_g::rt_trace_parity_pairs_g,            // 19 = kParityPairsGlobalVariant   shipped: tables beyond LDS
    parity_g::rt_trace_parity_g,                        // 20 = kParityGlobalVariant
};
constexpr int kParityCount = sizeof(kParityKernels) / sizeof(kParityKernels[0]);

int parity_variant_count() { return kParityCount; }
const char *parity_variant_name(int variant) {
    switch (variant) {
        case 0: return "rt_trace_parity";
        case kParityCoopVariant: return "rt_trace_parity_coop";
        case kParityW1Variant: return "rt_trace_parity_w1";
        case kParityCoopW1Variant: return "rt_trace_parity_coop_w1";
        case kParityPairsVariant: return "rt_trace_parity_pairs";
        case kParityPairsGlobalVariant: return "rt_trace_parity_pairs_g";
        case kParityGlobalVariant: return "rt_trace_parity_g";
        default: return "rt_trace_parity (a diagnostics instance)";
    }
}
int parity_variant_waves(int variant) { return (variant == kParityW1Variant || variant == kParityCoopW1Variant) ? 1 : 4; }

hipError_t launch_parity(int variant, const LaunchParams &p, dim3 grid, size_t lds, hipStream_t stream) {
    if (variant < 0 || variant >= kParityCount || !kParityKernels[variant]) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kParityKernels[variant], grid, dim3(64 * parity_variant_waves(variant)), lds, stream, p);
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
    for (KernelFn k : kParityKernels) {
        if (!k) continue;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace rt
