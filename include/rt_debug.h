/*
 * rt_debug.h -- diagnostics interface of librt_hip_diag.so (the library built with
 * -DRT_DIAGNOSTICS=1: every A/B and verification instance of the kernel, the exhaustive device-side
 * checks of the lean square root / reciprocal, scheduling knobs, device wall-clock logs).
 * NOT part of the drop-in boundary: librt_hip.so exports none of this.  Used by tests/ (parity of
 * every instance, scalar building blocks) and tools/ (A/B timing, stress, timelines).
 */
#ifndef RT_DEBUG_H
#define RT_DEBUG_H

#include "rt_api.h"

#ifdef __cplusplus
extern "C" {
#endif

/* rt_set_mode() of the diagnostics library also accepts 100+k / 200+k: row k of the parity / fast
 * instance tables (csrc/rt_kernel_parity.hip, rt_kernel_fast.hip; csrc/rt_device.h Instance).  Rows are
 * found by kernel symbol, never by number: rt_debug_instance("rt_trace_parity_coop_census") returns the
 * mode value (or RT_ERR_ARG), rt_debug_instance_name(fast, row) the symbol of a row ("" beyond the table). */
RT_API int rt_debug_variant_count(int fast);
RT_API int rt_debug_instance(const char *kernel_symbol);
RT_API const char *rt_debug_instance_name(int fast, int row);
/* the kernel instance the last launch of shard `shard` of a multi-device context used (rt_last_kernel names the first shard's) */
RT_API const char *rt_debug_shard_kernel(rt_ctx *ctx, int shard);
/* `repeats` launches of a kernel that does EXACTLY the table staging of the render kernels' prologue (every workgroup of the grid the
 * library would launch reads the scene tables into LDS) and nothing else: under rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum its counters are
 * the L2 hit rate of the LDS-staged sphere reads in isolation (tools/pmc_staging.sh) */
RT_API int rt_debug_stage_tables(rt_ctx *ctx, int n_samples, int repeats);
/* failure injection: puts a multi-device context into the state a failed gather (ncclGroupEnd) leaves it in and returns that
 * failure's code; every later rendering / state call on the context must then return RT_ERR_STATE until it is destroyed */
RT_API int rt_debug_break_gather(rt_ctx *ctx);
/* Bind `path` in the place of librccl.so.1 for multi-device contexts created AFTER this call (tests/rccl_double.cpp: sends
 * paired with receives as device-to-device copies on the streams it is handed; can be told to fail the k-th call of a
 * function), and -- second argument -- take a device listed n times as n devices, so that the grouped ncclRecv / ncclSend
 * branch (not the one-GPU rehearsal's copies) runs on a one-GPU box.  (NULL, 0) restores RCCL.  Contexts created before the
 * call keep their communicators of the previously bound library: destroy them first.                                    */
RT_API int rt_debug_set_rccl_library(const char *path, int repeated_devices_count_as_distinct);

/* Device-side evaluation of the scalar building blocks, for unit parity tests:
 * op 0: sinf, 1: cosf, 2: pow(x, 1/2.2f), 3: 1/x, 4: sqrt(x), 5: toInt(x) (result as float),
 * 6/7: the kernel's branch-free sinf/cosf (x >= 0), 8: the lean square root.                  */
RT_API int rt_debug_eval(int op, const float *in_host, float *out_host, size_t n);

/* lean sqrt against the compiler's sqrtf over all 2^32 inputs; sphere test with the unchecked root
 * against the one with sqrtf; candidate reciprocals per input exponent (out[4][256])          */
RT_API long long rt_debug_sqrt_mismatches(void);
RT_API long long rt_debug_hitpost_mismatches(void);
RT_API int rt_debug_rcp_probe(unsigned long long *out1024);

/* scheduling knobs (bit-invisible by test) */
RT_API int rt_debug_set_regen_gate(rt_ctx *ctx, int gate);        /* 0 = automatic, 1 = free-running      */
RT_API int rt_debug_set_mat_lds_limit(rt_ctx *ctx, int bytes);
RT_API int rt_debug_set_persist(rt_ctx *ctx, int on);
RT_API int rt_debug_set_ncus(rt_ctx *ctx, int n);
RT_API int rt_debug_set_coop_min(rt_ctx *ctx, int min_spheres);    /* the cooperative any-hit instances from this many spheres on (0 = never); a threshold set here decides alone -- the library's own measurement of coop against plain on scenes of 4 .. 11 spheres (rt_launch.hip launch_small) is switched off for the context */
/* hierarchy over the small spheres of large scenes: smallest tree that is built and used (0 = never; library default
 * 56), largest LDS footprint it is used at (0 = keep; default 31 KiB = five workgroups per CU).  rt_debug_read_bvh: the tables as the kernels
 * stage them (csrc/rt_device.h BvhTables) and counts4 = { always, leaves, stack depth, root pair } (slots = always + 8 * leaves), all 0 without a hierarchy */
RT_API int rt_debug_set_bvh(rt_ctx *ctx, int min_spheres, int lds_limit);
RT_API int rt_debug_set_tree_shape(rt_ctx *ctx, int by_area);        /* the hierarchy's shape: 1 (default) by surface area -- uploads below 1500 tree spheres on the host, larger ones and every device-resident update on the device; 2: on the device for every upload too; 0: the fixed (halved) shape everywhere */
RT_API int rt_debug_set_walk(rt_ctx *ctx, int steps, int gate, int forced);   /* steps: TAIL LANES since round 5, 0..63 -- a trip's walk phase ends once no more than `steps` lanes still walk, their walks go on in the next trip (0, the default: every walk runs to its end within the trip; an experiment's knob); gate: ready lanes that make a wavefront shade (0 = keep); forced 0 = hierarchy or plain sweep by estimate / measurement (default), 1 = the hierarchy whenever the scene has one */
RT_API int rt_debug_set_walk_round(rt_ctx *ctx, int steps);   /* pair steps a lane takes in a row before the leaf step of the lanes that hold a leaf (default 3; large = until every lane has one) */
/* n_rays rays { o.xyz, t_max, d.xyz, shadow != 0 } (8 floats each; the last as a bit pattern) through the hierarchy walk
 * AND the plain sweep, one lane per ray; out4 (4 words per ray) = the walk's answer, then the sweep's -- closest hit:
 * distance bits and scene index (~0 for a miss); shadow ray: the first blocking scene index (the sphere count for
 * none) and 0.  The two must be equal for every ray whatever its origin and direction. */
RT_API int rt_debug_walk_rays(rt_ctx *ctx, const float *rays8, uint32_t n_rays, uint32_t *out4);
RT_API int rt_debug_create_breakdown(double *out8);   /* host ms of the last rt_create of this process: device query, stream + events, allocations, kernel function attributes (code object load on a first context), seed stream generated, its upload, restore kernel + wait, total */
RT_API int rt_debug_tree_estimate(rt_ctx *ctx, double *out4);   /* { expected pair steps, expected leaf visits, predicted walk / sweep time per ray, verdict came from the estimate }; returns 1 if the scene has an estimate */
RT_API int rt_debug_set_choice_estimate(rt_ctx *ctx, int on);  /* 0: the surface-area estimate never decides hierarchy against sweep (every undecided scene is measured) */
RT_API int rt_debug_bvh_pick(rt_ctx *ctx);   /* 0 = not decided yet, 1 = the hierarchy, 2 = the plain sweep (of this scene, by measurement) */
RT_API int rt_debug_read_bvh(rt_ctx *ctx, float *blob_out, uint32_t cap_float4, uint32_t *counts4);
RT_API int rt_debug_read_packed_pairs(rt_ctx *ctx, void *out, uint32_t cap_bytes, uint32_t *n_pairs);   /* the packed pair table of the walk that reads its tables from HBM / L2 (csrc/rt_device.h BvhTables::packed_at): 32 bytes of frame { r0.xyz, - }, { scale.xyz, - }, then 32 bytes per pair in the pairs' order; *n_pairs = 0 when the scene's hierarchy has none */
RT_API int rt_debug_set_wg_waves(rt_ctx *ctx, int waves);          /* 0 = automatic, 1 or 4 wavefronts per workgroup */
RT_API int rt_debug_set_tile_order(rt_ctx *ctx, int on);           /* 0 = natural tile order; 1 = heavy first (the default) */
RT_API int rt_debug_read_tile_order(rt_ctx *ctx, uint32_t *order_out, uint32_t *cost_out, uint32_t cap, uint32_t *n_tiles, int *valid);

/* raw diagnostic counters (section census of the stamped instances; valid after rt_get_stats) */
RT_API int rt_debug_counters(rt_ctx *ctx, unsigned long long *out24);
RT_API int rt_debug_counters_raw(rt_ctx *ctx, unsigned long long *out32);

/* the reset this library used in round 1 (seed words restored by a copy on `hip_stream`, read back by
 * the next launch) and a probe kernel that counts seed words differing from the default stream into
 * counters[28] (probes run: [29]); both log their device wall-clock interval like the timelog instance */
RT_API int rt_debug_reset_by_copy(rt_ctx *ctx, void *hip_stream, int flags);   /* bit 0: hipMemcpyAsync instead of the copy kernel; bit 1: the kernel's waves end with an agent-scope release; bit 2: write-through (sc1) stores */
RT_API int rt_debug_probe_seeds(rt_ctx *ctx, void *hip_stream, int flags);     /* bit 0: the probe's waves start with an agent-scope acquire */
RT_API int rt_debug_sidelog_read(rt_ctx *ctx, uint32_t seq, unsigned long long *blocklog1024, uint32_t *stalelog64);

/* Device wall-clock log (s_memrealtime, 100 MHz): `entries` records of 8 u64 {first start, last end,
 * kind (1 render, 2 copy, 3 probe), user tag, stale words seen (probe), -, -, -}; every later launch of the timelog instance (mode 109) and every rt_debug_reset_by_copy /
 * rt_debug_probe_seeds on this context takes the next record.  rt_debug_timelog_tag sets the tag
 * stored with the following records (e.g. the frame number).  wave_entries > 0 also allocates a
 * per-wavefront log {start, end, xcc << 32 | HW_ID} for launches of mode 109.                   */
RT_API int rt_debug_timelog_enable(rt_ctx *ctx, uint32_t entries, uint32_t wave_entries);
RT_API int rt_debug_timelog_tag(rt_ctx *ctx, unsigned long long tag);
RT_API int rt_debug_timelog_read(rt_ctx *ctx, unsigned long long *out, uint32_t entries, uint32_t *used);
RT_API int rt_debug_wavelog_read(rt_ctx *ctx, unsigned long long *out, uint32_t wave_entries);

#ifdef __cplusplus
}
#endif
#endif /* RT_DEBUG_H */
