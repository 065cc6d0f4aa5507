// rt_debug.hip -- include/rt_debug.h: the diagnostics library's knobs, probes and exhaustive device-side checks (librt_hip_diag.so,
// -DRT_DIAGNOSTICS=1; tests/ and tools/ only).  Compiles to nothing in the product library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "rt_internal.h"

#ifndef RT_DIAGNOSTICS
#define RT_DIAGNOSTICS 0
#endif

using rt::fail;

#if RT_DIAGNOSTICS
using namespace rt;

// diagnostic only (rt_debug_stage_tables): EXACTLY the table staging of the render kernels' prologue -- every workgroup reads the
// geometry, light and (if they ride along) material tables into LDS -- and nothing else, so that the L2 counters of a profiler
// run show the hit rate of those reads in isolation (north_star: "L2-hit rate on the LDS-staged sphere reads").  One word per
// workgroup goes out so that the loads are not dead.
__global__ void rt_stage_probe_kernel(const rt::SceneTables T, int mat_in_lds, uint32_t *sink) {
    extern __shared__ float4 lds[];
    const uint32_t n = T.n_spheres, nl = T.n_lights;
    float4 *s_geom = lds, *s_la = s_geom + n, *s_lb = s_la + nl, *s_em = s_lb + nl, *s_co = s_em + n;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) s_geom[i] = T.geom[i];
    for (uint32_t i = threadIdx.x; i < nl; i += blockDim.x) {
        s_la[i] = T.lightA[i];
        s_lb[i] = T.lightB[i];
    }
    if (mat_in_lds)
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            s_em[i] = T.emis[i];
            s_co[i] = T.colr[i];
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        float acc = 0.f;
        for (uint32_t i = 0; i < n; ++i) acc += s_geom[i].w + (mat_in_lds ? s_em[i].x + s_co[i].x : 0.f);
        for (uint32_t i = 0; i < nl; ++i) acc += s_la[i].w + s_lb[i].w;
        sink[blockIdx.x & 1023u] = __float_as_uint(acc);
    }
}

// diagnostic only (rt_debug_reset_by_copy): the reset this library used in round 1 -- a copy kernel that
// restores the seed words, which the next launch then reads back.  Logs into its timelog record (tl) the
// device wall-clock of its first start / last end and the number of workgroups that ran, and per workgroup
// (blocklog) its start time and the XCD it ran on.  flags bit 1: every wave ends with an explicit
// agent-scope release (buffer_wbl2 sc1 + wait), i.e. the shader itself writes its XCD's L2 back instead
// of leaving that to the end-of-kernel action of the command processor.  flags bit 2: write-through stores; bit 3: atomic exchanges instead of stores.
__global__ void rt_debug_copy_seeds_kernel(uint32_t *seeds, const uint32_t *seeds0, size_t n, unsigned long long *tl,
                                           unsigned long long tag, unsigned long long *blocklog, int flags) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (tl && threadIdx.x == 0) atomicMin(&tl[0], t0);
    if (tl && threadIdx.x == 0 && blockIdx.x == 0) { tl[2] = 2ull; tl[3] = tag; }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (flags & 8) __hip_atomic_exchange(seeds + i, seeds0[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // memory-side read-modify-write
        else if (flags & 4) __hip_atomic_store(seeds + i, seeds0[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through (global_store ... sc1)
        else seeds[i] = seeds0[i];
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (flags & 2) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (threadIdx.x == 0) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (blocklog && (flags & 8)) __hip_atomic_exchange(&blocklog[blockIdx.x], (t0 << 4) | (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (blocklog) blocklog[blockIdx.x] = (t0 << 4) | (xcc & 15u);
        if (tl) atomicAdd(&tl[5], 1ull);                                    // workgroups that ran
        if (tl) atomicMax(&tl[1], __builtin_amdgcn_s_memrealtime());
    }
}

// flags bit 0: every wave starts with an explicit agent-scope acquire (buffer_inv sc1 + wait) before it reads
__global__ void rt_debug_probe_seeds_kernel(const uint32_t *seeds, const uint32_t *seeds0, size_t n, unsigned long long *out,
                                            unsigned long long *tl, unsigned long long tag, uint32_t *stalelog, int flags) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (tl && threadIdx.x == 0) atomicMin(&tl[0], t0);
    if (tl && threadIdx.x == 0 && blockIdx.x == 0) { tl[2] = 3ull; tl[3] = tag; }
    if (flags & 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (seeds[i] != seeds0[i]) {
            b += 1;
            if (stalelog) {                                                  // first 63 stale words: index | reader's XCD << 28
                const uint32_t k = atomicAdd(&stalelog[0], 1u);
                if (k < 63u) stalelog[1 + k] = (uint32_t)i | (xcc << 28);
            }
        }
    }
    if (b) atomicAdd(out, b);
    if (b && tl) atomicAdd(&tl[4], b);                                      // stale words seen by THIS probe
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(out + 1, 1ull);     // probes run
    if (tl && threadIdx.x == 0) atomicMax(&tl[1], __builtin_amdgcn_s_memrealtime());
}

extern "C" {

// =============================== diagnostics build only (rt_debug.h) ===========================

// host milliseconds of the last rt_create / rt_create_sharded of this process, by phase (g_create_ms above)
RT_API int rt_debug_create_breakdown(double *out8) {
    if (!out8) return fail(RT_ERR_ARG, "null argument");
    memcpy(out8, rt::create_breakdown(), 8 * sizeof(double));
    return RT_OK;
}
RT_API int rt_debug_variant_count(int fast) {
    int n = 0;
    (void)instances(fast != 0, &n);
    return n;
}
// the rt_set_mode value that selects the instance with this kernel symbol (100 + row / 200 + row), or RT_ERR_ARG
RT_API int rt_debug_instance(const char *name) {
    if (!name) return fail(RT_ERR_ARG, "name is null");
    for (int fast = 0; fast < 2; ++fast) {
        int n = 0;
        const rt::Instance *t = instances(fast != 0, &n);
        for (int k = 0; k < n; ++k)
            if (strcmp(t[k].name, name) == 0) return (fast ? 200 : 100) + k;
    }
    return fail(RT_ERR_ARG, "no instance named %s in this library", name);
}
// the kernel instance the last launch of shard `shard` of a multi-device context used ("" for a plain context or beyond the shards)
RT_API const char *rt_debug_shard_kernel(rt_ctx *c, int shard) {
    if (!c || !c->multi || shard < 0 || shard >= rt::multi_shards(c)) return "";
    return rt::multi_shard(c, shard)->last_kernel;
}
// (readers of the tables and the hierarchy: device-resident updates leave them stale until something asks -- rt_scene.hip refresh_tables)
static int fresh_tables(rt_ctx *c) { return c->tables_stale ? rt::refresh_tables(c, c->stream) : RT_OK; }

// The render kernels' table staging alone (rt_stage_probe_kernel), `repeats` launches of the grid and workgroup shape the library
// would use for `n_samples` passes of the current scene: for a profiler run that isolates the L2 behaviour of those reads.
RT_API int rt_debug_stage_tables(rt_ctx *c, int n_samples, int repeats) {
    if (!c || c->multi || !c->have_scene) return fail(RT_ERR_ARG, "null / multi-device context, or no scene");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = fresh_tables(c);
    if (rc != RT_OK) return rc;
    rc = chain(c, c->stream);
    if (rc != RT_OK) return rc;
    const size_t lds_all = rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, true, n_samples);
    const int mat = lds_all <= (size_t)c->mat_lds_limit ? 1 : 0;
    const size_t lds = rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, mat != 0, n_samples);
    if (lds > kLdsMax) return fail(RT_ERR_ARG, "tables of %zu B do not fit LDS", lds);
    const bool coop = c->coop_min > 0 && c->scene.n_spheres >= (uint32_t)c->coop_min;
    const bool w1 = lds + (coop ? 1536u : 256u) <= 6 * 1024;
    const int tile_w = w1 ? 8 : 32;
    const dim3 grid((unsigned)((c->w + tile_w - 1) / tile_w), (unsigned)((c->local_rows + 7) / 8));
    uint32_t *sink = reinterpret_cast<uint32_t *>(c->d_tile_cost);          // (scratch: n_tiles >= 1024 words are not needed -- index & 1023 of a buffer that large)
    if (!sink || c->n_tiles < 1024) return fail(RT_ERR_ARG, "image too small for the probe");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rt_stage_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_debug_stage_tables: %s", hipGetErrorString(e));
    for (int k = 0; k < repeats; ++k) {
        hipLaunchKernelGGL(rt_stage_probe_kernel, grid, dim3(w1 ? 64 : 256), lds, c->stream, c->scene, mat, sink);
        HIP_TRY(hipGetLastError());
    }
    c->cost_valid = c->order_valid = false;                                 // (the probe scribbled over the tile costs)
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}
// failure injection: the state a failed gather (ncclGroupEnd) leaves a multi-device context in -- every later call is refused
RT_API int rt_debug_break_gather(rt_ctx *c) {
    if (!c || !c->multi) return fail(RT_ERR_ARG, "not a multi-device context");
    return rt::multi_debug_break(c);
}
// test double in the place of RCCL (tests/rccl_double.cpp), and a repeated device list taken as distinct devices: the grouped
// ncclRecv / ncclSend branch of rt_multi.hip and its failure handling then run on one GPU.  path = NULL, 0 restores RCCL.
RT_API int rt_debug_set_rccl_library(const char *path, int repeated_devices_count_as_distinct) {
    return rt::multi_debug_set_rccl(path, repeated_devices_count_as_distinct);
}
// the kernel symbol of row `row` of the parity (fast = 0) or fast table, or "" beyond it
RT_API const char *rt_debug_instance_name(int fast, int row) {
    int n = 0;
    const rt::Instance *t = instances(fast != 0, &n);
    return (row >= 0 && row < n) ? t[row].name : "";
}

static int dbg_set_gate(rt_ctx *c, int v) { c->regen_gate = v; return RT_OK; }
static int dbg_set_matlds(rt_ctx *c, int v) { c->mat_lds_limit = v; return RT_OK; }
static int dbg_set_persist(rt_ctx *c, int v) { c->persist = v ? 1 : 0; return RT_OK; }
static int dbg_set_ncus(rt_ctx *c, int v) { c->n_cus = v; return RT_OK; }
static int dbg_set_coop(rt_ctx *c, int v) { c->coop_min = v & 0xffffff; c->coop_kmax = v >> 24; c->coop_probe = 0; rearm_probe(c); return RT_OK; }     // (a threshold set by hand decides alone: no measurement)
static int dbg_set_wg(rt_ctx *c, int v) { c->wg_waves = v; return RT_OK; }
static int dbg_set_order(rt_ctx *c, int v) { c->use_order = v ? 1 : 0; c->order_valid = false; return RT_OK; }
static int dbg_apply(rt_ctx *c, int (*fn)(rt_ctx *, int), int v) { return c->multi ? rt::multi_debug_each(c, fn, v) : fn(c, v); }

// tuning knob (not part of the contract): 0 = automatic, 1 = free-running, n = gate of n lanes
RT_API int rt_debug_set_regen_gate(rt_ctx *c, int gate) {
    if (!c || gate < 0 || gate > 64) return fail(RT_ERR_ARG, "gate %d", gate);
    return dbg_apply(c, dbg_set_gate, gate);
}
RT_API int rt_debug_set_mat_lds_limit(rt_ctx *c, int bytes) {
    if (!c || bytes < 0) return fail(RT_ERR_ARG, "bytes %d", bytes);
    return dbg_apply(c, dbg_set_matlds, bytes);
}
RT_API int rt_debug_set_persist(rt_ctx *c, int on) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    return dbg_apply(c, dbg_set_persist, on);
}
RT_API int rt_debug_set_ncus(rt_ctx *c, int n) {      // shrink the persistent grid (tests of the tile queue)
    if (!c || n < 1) return fail(RT_ERR_ARG, "n %d", n);
    return dbg_apply(c, dbg_set_ncus, n);
}
RT_API int rt_debug_set_wg_waves(rt_ctx *c, int waves) {    // 0 = automatic, 1 = single-wavefront workgroups, 4 = four wavefronts
    if (!c || (waves != 0 && waves != 1 && waves != 4)) return fail(RT_ERR_ARG, "waves %d", waves);
    return dbg_apply(c, dbg_set_wg, waves);
}
RT_API int rt_debug_set_tile_order(rt_ctx *c, int on) {      // 0: tiles in their natural order (the round-1 behaviour); 1: heavy first
    if (!c || on < 0 || on > 1) return fail(RT_ERR_ARG, "ctx is null / order %d", on);
    return dbg_apply(c, dbg_set_order, on);
}
// the tile order in use (valid = 0: none, tiles run in their natural order) and the per-tile costs of the last launch
RT_API int rt_debug_read_tile_order(rt_ctx *c, uint32_t *order_out, uint32_t *cost_out, uint32_t cap, uint32_t *n_tiles, int *valid) {
    if (!c || c->multi) return fail(RT_ERR_ARG, "null / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    const uint32_t in_use = c->cost_valid && c->cost_tiles ? c->cost_tiles : c->n_tiles;      // (the tiles of the last launch's workgroup shape)
    const uint32_t n = cap < in_use ? cap : in_use;
    if (order_out && n) HIP_TRY(hipMemcpy(order_out, c->d_order, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (cost_out && n) HIP_TRY(hipMemcpy(cost_out, c->d_tile_cost, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (n_tiles) *n_tiles = in_use;
    if (valid) *valid = c->order_valid ? 1 : 0;
    return RT_OK;
}
// the hierarchy of large scenes: min_spheres = smallest tree that is built and used (0 = never), lds_limit = largest
// LDS footprint it is used at (0 = keep).  Takes effect at once: the current scene's tables are rebuilt.
static int dbg_set_bvh_lds(rt_ctx *c, int v) { if (v > 0) c->bvh_lds_limit = v; return RT_OK; }
static int dbg_set_bvh_min(rt_ctx *c, int v) {
    c->bvh_min = v;
    rearm_probe(c);
    if (!c->have_scene) return RT_OK;
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = fresh_tables(c);
    if (rc != RT_OK) return rc;
    rc = chain(c, c->stream);
    return rc != RT_OK ? rc : rt::build_bvh(c, c->scene.n_spheres, c->stream, true);
}
static int dbg_set_tree_shape(rt_ctx *c, int v) {
    c->bvh_sah = v < 0 ? 0 : (v > 2 ? 2 : v);
    return dbg_set_bvh_min(c, c->bvh_min);          // (rebuilds the current scene's tables, re-arms the probe)
}
// The hierarchy's shape.  1 (default): by surface area -- uploads whose choice of form is estimated (below 1500 tree spheres) on the host,
// larger uploads and every device-resident update on the device; 2: on the device for every upload too; 0: the fixed (halved) shape
// everywhere.  Takes effect at once.
RT_API int rt_debug_set_tree_shape(rt_ctx *c, int by_area) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    return dbg_apply(c, dbg_set_tree_shape, by_area);
}
static int dbg_set_walk_gate(rt_ctx *c, int v) { if (v > 0) c->walk_gate = v; return RT_OK; }
static int dbg_set_walk_round(rt_ctx *c, int v) { c->walk_round = v; return RT_OK; }
static int dbg_set_walk_tail(rt_ctx *c, int v) { c->walk_tail = v; return RT_OK; }

RT_API int rt_debug_set_walk_round(rt_ctx *c, int steps) {
    if (!c || steps < 1) return fail(RT_ERR_ARG, "steps %d", steps);
    return dbg_apply(c, dbg_set_walk_round, steps);
}

static int dbg_set_walk_forced(rt_ctx *c, int v) { c->walk_forced = v ? 1 : 0; rearm_probe(c); return RT_OK; }
// rt_walk.inc.h: pair steps per lane per loop trip, ready lanes that make a wavefront shade (0 = keep either), and
// forced: 0 = hierarchy or plain sweep by measurement (the library's behaviour), 1 = the hierarchy whenever the scene has one
RT_API int rt_debug_set_walk(rt_ctx *c, int steps, int gate, int forced) {
    // `steps` = TAIL LANES since round 5 (rounds 2-3: the per-trip step budget): the walk phase of a trip ends once no more than `steps` lanes
    // still walk (0 = every walk runs to its end within the trip) -- it rides in the high bits of walk_round
    if (!c || steps < 0 || steps > 63 || gate < 0 || gate > 64 || forced < 0 || forced > 1)
        return fail(RT_ERR_ARG, "tail lanes %d (0..63), gate %d (0..64), forced %d (0 / 1)", steps, gate, forced);
    int rc = dbg_apply(c, dbg_set_walk_tail, steps);
    if (rc != RT_OK) return rc;
    rc = dbg_apply(c, dbg_set_walk_gate, gate);
    return rc != RT_OK ? rc : dbg_apply(c, dbg_set_walk_forced, forced);
}
// rays8[i] = { o.xyz, t_max, d.xyz, shadow != 0 } through the hierarchy walk and through the plain sweep (csrc/rt_walk.inc.h
// rt_walk_rays kernel); out4[i] = the walk's answer, then the sweep's (closest: distance bits, scene index; shadow: first
// blocking index, 0)
RT_API int rt_debug_walk_rays(rt_ctx *c, const float *rays8, uint32_t n_rays, uint32_t *out4) {
    if (!c || c->multi || !rays8 || !out4) return fail(RT_ERR_ARG, "null / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = fresh_tables(c);
    if (rc != RT_OK) return rc;
    if (!c->bvh_ok) return fail(RT_ERR_STATE, "the scene has no hierarchy (rt_debug_set_bvh)");
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    if (!c->have_cam) c->cam = rt_camera{};
    rt::LaunchParams p = make_params(c, 1);
    p.bvh = c->bvh;
    const size_t lds = rt::lds_bytes_pairs(0, 0, false, 0, c->bvh.n_leaves, c->bvh.n_slots, c->bvh.stack_depth, 256);
    if (lds > 152 * 1024) return fail(RT_ERR_ARG, "tables need %zu B of LDS", lds);
    float4 *d_rays = nullptr;
    uint4 *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_rays, (size_t)n_rays * 32 + 32));
    hipError_t e = hipMalloc(&d_out, (size_t)n_rays * 16 + 16);
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays8, (size_t)n_rays * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rt::launch_walk_rays(p, d_rays, n_rays, d_out, lds, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out4, d_out, (size_t)n_rays * 16, hipMemcpyDeviceToHost);
    (void)hipFree(d_rays);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_debug_walk_rays: %s", hipGetErrorString(e));
    return RT_OK;
}
// the estimate that settles hierarchy against sweep without a launch: out4 = { expected pair steps, expected leaf visits,
// predicted walk / sweep time per ray, 1 if the verdict in force came from it (0: measured, or none yet) }; returns 1 when
// the context holds an estimate for its scene.  rt_debug_set_choice_estimate(ctx, 0) switches it off: every undecided
// scene is then measured (the calibration's way of getting both timings).
RT_API int rt_debug_tree_estimate(rt_ctx *c, double *out4) {
    if (!c || c->multi || !out4) return fail(RT_ERR_ARG, "null / multi-device context");
    if (c->tables_stale && select_device(c) == RT_OK) (void)fresh_tables(c);
    out4[0] = c->bvh_est_pairs;
    out4[1] = c->bvh_est_leaves;
    out4[2] = c->bvh_est_valid && c->bvh_ok ? estimate_ratio(c) : 0.0;
    out4[3] = c->pick_estimated ? 1.0 : 0.0;
    return c->bvh_est_valid && c->bvh_ok ? 1 : 0;
}
static int dbg_set_estimate(rt_ctx *c, int v) { c->use_estimate = v ? 1 : 0; rearm_probe(c); return RT_OK; }
RT_API int rt_debug_set_choice_estimate(rt_ctx *c, int on) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    return dbg_apply(c, dbg_set_estimate, on);
}
RT_API int rt_debug_bvh_pick(rt_ctx *c) {
    if (!c || c->multi) return fail(RT_ERR_ARG, "null / multi-device context");
    if (select_device(c) == RT_OK) probe_poll(c, false);
    return c->bvh_pick;
}
RT_API int rt_debug_set_bvh(rt_ctx *c, int min_spheres, int lds_limit) {
    if (!c || min_spheres < 0 || lds_limit < 0 || lds_limit > 152 * 1024) return fail(RT_ERR_ARG, "min_spheres %d, lds_limit %d", min_spheres, lds_limit);
    int rc = dbg_apply(c, dbg_set_bvh_lds, lds_limit);
    return rc != RT_OK ? rc : dbg_apply(c, dbg_set_bvh_min, min_spheres);
}
// the packed pair table (rt_device.h BvhTables::packed_at) as it lies in HBM: 2 float4 of frame, then 32 bytes per pair; *n_pairs = 0 without one
RT_API int rt_debug_read_packed_pairs(rt_ctx *c, void *out, uint32_t cap_bytes, uint32_t *n_pairs) {
    if (!c || c->multi || !n_pairs) return fail(RT_ERR_ARG, "null / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = fresh_tables(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    *n_pairs = 0;
    if (!c->bvh_ok || c->bvh.packed_at == 0 || c->bvh.n_leaves < 2) return RT_OK;
    const size_t need = 32 + 32 * (size_t)(c->bvh.n_leaves - 1);
    *n_pairs = c->bvh.n_leaves - 1;
    if (out) {
        if (cap_bytes < need) return fail(RT_ERR_ARG, "the packed table takes %zu bytes", need);
        HIP_TRY(hipMemcpy(out, c->d_bvh + c->bvh.packed_at, need, hipMemcpyDeviceToHost));
    }
    return RT_OK;
}
// the blob of rt_device.h BvhTables as it lies in HBM (float4 units), and four numbers {always, leaves, stack depth, root pair}
// (slots = always + 8 * leaves);
// counts of 0 = the scene has no hierarchy
RT_API int rt_debug_read_bvh(rt_ctx *c, float *blob_out, uint32_t cap_float4, uint32_t *counts4) {
    if (!c || c->multi || !counts4) return fail(RT_ERR_ARG, "null / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = fresh_tables(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    counts4[0] = counts4[1] = counts4[2] = counts4[3] = 0;
    if (!c->bvh_ok) return RT_OK;
    float4 hdr[2];                          // (the root pair travels in the header: a tree shaped on the device has it where only the device knows)
    HIP_TRY(hipMemcpy(hdr, c->d_bvh, sizeof hdr, hipMemcpyDeviceToHost));
    uint32_t last;
    memcpy(&last, &hdr[1].w, 4);
    counts4[0] = c->bvh.n_always; counts4[1] = c->bvh.n_leaves; counts4[2] = c->bvh.stack_depth; counts4[3] = last & 0xffffu;
    const size_t need = rt::bvh_blob_float4s(c->bvh.n_leaves, c->bvh.n_slots);
    if (blob_out) {
        if (cap_float4 < need) return fail(RT_ERR_ARG, "blob needs %zu float4", need);
        HIP_TRY(hipMemcpy(blob_out, c->d_bvh, need * sizeof(float4), hipMemcpyDeviceToHost));
    }
    return RT_OK;
}
// min_spheres | kmax << 24: scenes of at least min_spheres use the cooperative any-hit instance; kmax: it shares a sweep out only while
// no more than kmax shadow rays are pending in the wavefront (0 = no limit)
RT_API int rt_debug_set_coop_min(rt_ctx *c, int min_spheres) {
    if (!c || min_spheres < 0) return fail(RT_ERR_ARG, "min_spheres %d", min_spheres);
    return dbg_apply(c, dbg_set_coop, min_spheres);
}

// section cycle sums of a stamped instance (valid after rt_get_stats)
RT_API int rt_debug_counters(rt_ctx *c, unsigned long long *out24) {
    if (!c || !out24 || c->multi) return fail(RT_ERR_ARG, "null argument / multi-device context");
    memcpy(out24, c->debug_counters, sizeof c->debug_counters);
    return RT_OK;
}

RT_API int rt_debug_counters_raw(rt_ctx *c, unsigned long long *out32) {
    if (!c || !out32 || c->multi) return fail(RT_ERR_ARG, "null argument / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = chain(c, c->stream);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out32, c->d_counters, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

static unsigned long long *timelog_next(rt_ctx *c, uint32_t *seq_out) {
    if (!c->d_timelog || c->timelog_used >= c->timelog_cap) return nullptr;
    *seq_out = c->timelog_used;
    return c->d_timelog + 8 * (size_t)(c->timelog_used++);
}

// NOTE: deliberately NOT chained (no event dependency added by the library): this is the round-1 reset,
// kept to reproduce and study the ordering failure recorded in DESIGN.md section 3
RT_API int rt_debug_reset_by_copy(rt_ctx *c, void *hip_stream, int flags) {
    if (!c || c->multi) return fail(RT_ERR_ARG, "null / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    const size_t n = 2 * (size_t)c->w * (size_t)c->h;
    if (flags & 1) {
        HIP_TRY(hipMemcpyAsync(c->d_seeds, c->d_seeds0, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
    } else {
        uint32_t seq = 0;
        unsigned long long *tl = timelog_next(c, &seq);
        unsigned long long *bl = (tl && c->d_blocklog) ? c->d_blocklog + (size_t)seq * 1024 : nullptr;
        hipLaunchKernelGGL(rt_debug_copy_seeds_kernel, dim3(1024), dim3(256), 0, (hipStream_t)hip_stream, c->d_seeds, c->d_seeds0, n, tl,
                           c->timelog_tag, bl, flags);
        HIP_TRY(hipGetLastError());
    }
    c->seeds_default = false;
    c->current_sample = 0;
    return RT_OK;
}

// a kernel on `hip_stream` that counts the seed words differing from the default stream
// into counters[28] (and the number of probes into counters[29]); read them with rt_debug_counters_raw
RT_API int rt_debug_probe_seeds(rt_ctx *c, void *hip_stream, int flags) {
    if (!c || c->multi) return fail(RT_ERR_ARG, "null / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    uint32_t seq = 0;
    unsigned long long *tl = timelog_next(c, &seq);
    uint32_t *sl = (tl && c->d_stalelog) ? c->d_stalelog + (size_t)seq * 64 : nullptr;
    hipLaunchKernelGGL(rt_debug_probe_seeds_kernel, dim3(256), dim3(256), 0, (hipStream_t)hip_stream, c->d_seeds, c->d_seeds0,
                       2 * (size_t)c->w * (size_t)c->h, c->d_counters + 28, tl, c->timelog_tag, sl, flags);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

RT_API int rt_debug_timelog_enable(rt_ctx *c, uint32_t entries, uint32_t wave_entries) {
    if (!c || c->multi) return fail(RT_ERR_ARG, "null / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    (void)hipFree(c->d_timelog);
    (void)hipFree(c->d_wavelog);
    (void)hipFree(c->d_blocklog);
    (void)hipFree(c->d_stalelog);
    c->d_timelog = c->d_wavelog = c->d_blocklog = nullptr;
    c->d_stalelog = nullptr;
    c->timelog_cap = c->timelog_used = c->wavelog_cap = 0;
    if (entries) {
        HIP_TRY(hipMalloc(&c->d_blocklog, (size_t)entries * 1024 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(c->d_blocklog, 0, (size_t)entries * 1024 * sizeof(unsigned long long)));
        if (wave_entries == 0xC0FFEEu) {
            // provenance experiment (tools/gather_stress.py RT_LOG_PATTERN=1): the zeros above were written by the
            // runtime's fill KERNEL (shader stores through some XCD's L2); now the same bytes are overwritten with a
            // pattern by a host-to-device copy that does not go through any L2.  A workgroup log entry that is later
            // found lost then tells by its value what happened: the pattern = the write never arrived; zero = a stale
            // line from the fill kernel was written back over it afterwards.
            HIP_TRY(hipDeviceSynchronize());
            std::vector<unsigned long long> pat((size_t)entries * 1024, 0x5555555555555550ull);
            HIP_TRY(hipMemcpy(c->d_blocklog, pat.data(), pat.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
            HIP_TRY(hipDeviceSynchronize());
            wave_entries = 0;
        }
        HIP_TRY(hipMalloc(&c->d_stalelog, (size_t)entries * 64 * sizeof(uint32_t)));
        HIP_TRY(hipMemset(c->d_stalelog, 0, (size_t)entries * 64 * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->d_timelog, (size_t)entries * 8 * sizeof(unsigned long long)));
        std::vector<unsigned long long> init((size_t)entries * 8, 0ull);
        for (uint32_t i = 0; i < entries; ++i) init[8 * (size_t)i] = ~0ull;
        HIP_TRY(hipMemcpy(c->d_timelog, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        c->timelog_cap = entries;
    }
    if (wave_entries) {
        HIP_TRY(hipMalloc(&c->d_wavelog, (size_t)wave_entries * 3 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(c->d_wavelog, 0, (size_t)wave_entries * 3 * sizeof(unsigned long long)));
        c->wavelog_cap = wave_entries;
    }
    return RT_OK;
}

RT_API int rt_debug_timelog_tag(rt_ctx *c, unsigned long long tag) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    c->timelog_tag = tag;
    return RT_OK;
}

RT_API int rt_debug_timelog_read(rt_ctx *c, unsigned long long *out, uint32_t entries, uint32_t *used) {
    if (!c || !out || c->multi) return fail(RT_ERR_ARG, "null argument / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    const uint32_t n = entries < c->timelog_cap ? entries : c->timelog_cap;
    if (n) HIP_TRY(hipMemcpy(out, c->d_timelog, (size_t)n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (used) *used = c->timelog_used;
    return RT_OK;
}

// the per-workgroup log of copy record `seq` (1024 u64: start << 4 | xcc) and the stale-word log of probe
// record `seq` (64 u32: count, then index | reader xcc << 28)
RT_API int rt_debug_sidelog_read(rt_ctx *c, uint32_t seq, unsigned long long *blocklog1024, uint32_t *stalelog64) {
    if (!c || c->multi || seq >= c->timelog_cap) return fail(RT_ERR_ARG, "null / multi-device context / record %u", seq);
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    if (blocklog1024) HIP_TRY(hipMemcpy(blocklog1024, c->d_blocklog + (size_t)seq * 1024, 1024 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (stalelog64) HIP_TRY(hipMemcpy(stalelog64, c->d_stalelog + (size_t)seq * 64, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

RT_API int rt_debug_wavelog_read(rt_ctx *c, unsigned long long *out, uint32_t wave_entries) {
    if (!c || !out || c->multi) return fail(RT_ERR_ARG, "null argument / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    const uint32_t n = wave_entries < c->wavelog_cap ? wave_entries : c->wavelog_cap;
    if (n) HIP_TRY(hipMemcpy(out, c->d_wavelog, (size_t)n * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RT_OK;
}

// exhaustive device-side check of the lean correctly-rounded sqrt: mismatches over all 2^32 inputs
static long long sqrt_check(int which) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device");
    unsigned long long *d = nullptr, h = 0;
    if (hipSetDevice(0) != hipSuccess || hipMalloc(&d, 8) != hipSuccess) return fail(RT_ERR_HIP, "alloc");
    hipError_t e = hipMemset(d, 0, 8);
    if (e == hipSuccess) e = rt::launch_sqrt_check(d, nullptr, which);
    if (e == hipSuccess) e = hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "sqrt check %d: %s", which, hipGetErrorString(e));
    return (long long)h;
}
RT_API long long rt_debug_sqrt_mismatches(void) { return sqrt_check(0); }
// sphere test with the unchecked square root against the one with sqrtf, tiny discriminants
RT_API long long rt_debug_hitpost_mismatches(void) { return sqrt_check(1); }

// mismatches of the candidate lean reciprocals per input exponent: out[4][256]
RT_API int rt_debug_rcp_probe(unsigned long long *out1024) {
    if (!out1024) return fail(RT_ERR_ARG, "null argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device");
    unsigned long long *d = nullptr;
    if (hipSetDevice(0) != hipSuccess || hipMalloc(&d, 8192) != hipSuccess) return fail(RT_ERR_HIP, "alloc");
    hipError_t e = hipMemset(d, 0, 8192);
    if (e == hipSuccess) e = rt::launch_rcp_probe(d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out1024, d, 8192, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_debug_rcp_probe: %s", hipGetErrorString(e));
    return RT_OK;
}

RT_API int rt_debug_eval(int op, const float *in_host, float *out_host, size_t n) {
    if ((!in_host || !out_host) && n) return fail(RT_ERR_ARG, "null argument");
    if (op < 0 || op > 8) return fail(RT_ERR_ARG, "op %d", op);
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device");
    if (n == 0) return RT_OK;
    float *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(hipSetDevice(0));
    HIP_TRY(hipMalloc(&d_in, n * sizeof(float)));
    hipError_t e = hipMalloc(&d_out, n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_in, in_host, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rt::launch_eval_parity(op, d_in, d_out, n, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_host, d_out, n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_debug_eval: %s", hipGetErrorString(e));
    return RT_OK;
}

}  // extern "C"
#endif   // RT_DIAGNOSTICS
