// rt_launch.hip -- what one launch of the render kernel is made of: which kernel instance (rt_device.h Instance: arithmetic mode x
// role x workgroup shape), its tables and LDS, the scheduling data it runs with (heavy-first tile order) and, for scenes
// that have a hierarchy, whether it is walked or the plain sweep runs -- settled by a
// surface-area estimate at rt_set_scene, by measurement inside the estimate's band.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_internal.h"

#ifndef RT_DIAGNOSTICS
#define RT_DIAGNOSTICS 0
#endif

using rt::fail;

// Heavy-first order of the tiles from the costs the last launch left (rt_trace.inc.h): ONE workgroup; a counting sort over
// 1024 cost classes (largest first; the order inside a class does not matter).
__global__ void __launch_bounds__(1024) rt_order_tiles_kernel(const uint32_t *cost, uint32_t *order, uint32_t n) {
    __shared__ unsigned s_max;
    __shared__ unsigned s_hist[1024];
    const unsigned tid = threadIdx.x;
    constexpr unsigned kCap = 0x1FFFFFu;            // 21 ms of ticks: cost * 1023 stays inside 32 bits
    auto key_of = [&](uint32_t i) -> unsigned { return cost[i] < kCap ? cost[i] : kCap; };
    if (tid == 0) s_max = 1u;
    s_hist[tid] = 0u;
    __syncthreads();
    unsigned m = 0;
    for (uint32_t i = tid; i < n; i += 1024) {
        const unsigned c_ = key_of(i);
        m = c_ > m ? c_ : m;
    }
    atomicMax(&s_max, m);
    __syncthreads();
    const unsigned top = s_max;
    for (uint32_t i = tid; i < n; i += 1024) atomicAdd(&s_hist[1023u - key_of(i) * 1023u / top], 1u);
    __syncthreads();
    if (tid == 0) {                     // exclusive prefix over the classes, most expensive class first
        unsigned run = 0;
        for (int k = 0; k < 1024; ++k) {
            const unsigned c_ = s_hist[k];
            s_hist[k] = run;
            run += c_;
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 1024) order[atomicAdd(&s_hist[1023u - key_of(i) * 1023u / top], 1u)] = i;
}

namespace rt {

rt::LaunchParams make_params(rt_ctx *c, int n_samples) {
    rt::LaunchParams p{};
    p.scene = c->scene;
    p.cam = c->cam;
    p.seeds = c->d_seeds;
    p.seeds_in = c->seeds_default ? c->d_seeds0 : c->d_seeds;
    p.colors = c->d_colors;
    p.pixels = c->d_pixels_ext ? c->d_pixels_ext : c->d_pixels;
    p.counters = c->d_counters;
    p.stats = c->d_stats;
    p.w = c->w;
    p.h = c->h;
    p.first_sample = c->current_sample;
    p.n_samples = n_samples;
    p.rank = c->rank;
    p.nranks = c->nranks;
    p.tile_rows = c->tile_rows;
    p.local_rows = c->local_rows;
    p.skip_pixels = c->pixel_write ? 0 : 1;
    p.inv_w = 1.f / (float)c->w;          // correctly rounded on the host as on the device (-ffp-contract=off, IEEE division)
    p.inv_h = 1.f / (float)c->h;
    p.regen_gate = c->regen_gate > 0 ? c->regen_gate : (c->scene.n_spheres <= 512 ? 8 : 1);
    p.coop_kmax = c->coop_kmax;
    p.tiles_x = (c->w + 7) / 8;
    p.n_tiles = p.tiles_x * ((c->local_rows + 7) / 8);
    return p;
}


// ---- which instance, and what it needs (rt_device.h Instance; the rows live next to the instantiations) ----

const rt::Instance *instances(bool fast, int *count) { return fast ? rt::fast_instances(count) : rt::parity_instances(count); }

// the row with this role and workgroup shape (null: this library has none)
static const rt::Instance *find_role(bool fast, int role, int waves) {
    int n = 0;
    const rt::Instance *t = instances(fast, &n);
    for (int k = 0; k < n; ++k)
        if (t[k].role == role && t[k].waves == waves) return &t[k];
    return nullptr;
}

// LDS the hierarchy's staged tables take for this scene
static size_t pairs_lds(const rt_ctx *c, bool mat, int n_samples, int waves = 4) {
    return rt::lds_bytes_pairs(c->scene.n_spheres, c->scene.n_lights, mat, n_samples, c->bvh.n_leaves, c->bvh.n_slots, c->bvh.stack_depth, 64 * waves);
}

// the plain sweep's tables (geometry and lights) fit LDS for this launch
bool tables_fit_lds(const rt_ctx *c, int n_samples) {
    return rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, false, n_samples) <= kLdsMax;
}
// ... and are staged there: while at least four workgroups of that size fit a CU.  Larger tables go through the scalar cache (rt_trace_*_g), which
// keeps six wavefronts per SIMD at any size: a sweep over 96 KB of staged tables runs ONE workgroup per CU and takes five times as long (rt_internal.h)
static bool sweep_stages_tables(const rt_ctx *c, int n_samples) {
    return rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, false, n_samples) <= (size_t)(c->sweep_lds_limit < (int)kLdsMax ? c->sweep_lds_limit : (int)kLdsMax);
}

// the hierarchy's tables fit the LDS budget given to them; otherwise the walk reads them from HBM / L2
static bool bvh_fits_lds(const rt_ctx *c, int n_samples) { return pairs_lds(c, false, n_samples) <= (size_t)c->bvh_lds_limit; }
// ... or at least its PAIRS do (header | pairs | stacks): the chain of dependent fetches a walk consists of then stays in LDS
static size_t pairs_only_lds(const rt_ctx *c, int n_samples, int waves = 4) {
    return rt::lds_bytes_pairs(0, 0, false, n_samples, c->bvh.n_leaves, 0, c->bvh.stack_depth, 64 * waves);
}
static bool bvh_pairs_fit_lds(const rt_ctx *c, int n_samples) { return c->bvh_mixed != 0 && pairs_only_lds(c, n_samples) <= (size_t)c->bvh_lds_limit; }

// the scene has a hierarchy and the context may use it
static bool bvh_usable(const rt_ctx *c) { return c->bvh_ok && c->wg_waves != 1 && c->persist == 0; }

// What the instance needs from the context, checked against what the context has: the ONE place that sizes the
// dynamic LDS and hands out the hierarchy.  An instance whose tables the context lacks is refused (RT_ERR_STATE),
// whatever route selected it -- the measured choice, a forced form, or a diagnostics mode.
static int bind_tables(rt_ctx *c, const rt::Instance &inst, int n_samples, rt::LaunchParams &p, size_t *lds_out) {
    const bool needs_bvh = inst.tables == rt::kTabPairsLds || inst.tables == rt::kTabPairsGlobal || inst.tables == rt::kTabPairsLdsSlotsGlobal ||
                           inst.tables == rt::kTabPairsTopLds || inst.tables == rt::kTabPairsPacked;
    p.bvh = rt::BvhTables{};
    if (needs_bvh) {
        if (!c->bvh_ok || !c->bvh.blob)
            return fail(RT_ERR_STATE, "%s walks a hierarchy and the scene has none (fewer than %d small spheres?)", inst.name, c->bvh_min);
        p.bvh = c->bvh;
    }
    size_t lds = 0;
    switch (inst.tables) {
        case rt::kTabSweepLds:
            lds = rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, p.mat_in_lds != 0, n_samples);
            break;
        case rt::kTabSweepGlobal:
            p.mat_in_lds = 0;
            lds = rt::lds_bytes(0, 0, false, n_samples);
            break;
        case rt::kTabPairsLds:
            p.mat_in_lds = 0;               // (the walk reads a hit's material by slot from the hierarchy's blob: nothing of it is staged)
            lds = pairs_lds(c, false, n_samples, inst.waves * ((inst.flags & rt::kInstTwoRays) ? 2 : 1));
            break;
        case rt::kTabPairsGlobal:
            p.mat_in_lds = 0;
            lds = rt::lds_bytes_pairs(0, 0, false, n_samples, 1, 0, c->bvh.stack_depth, 64 * inst.waves);
            break;
        case rt::kTabPairsLdsSlotsGlobal:
            p.mat_in_lds = 0;
            lds = pairs_only_lds(c, n_samples, inst.waves);
            break;
        case rt::kTabPairsPacked:           // header | the packed table's frame | stacks
            if (c->bvh.packed_at == 0 && c->bvh.n_leaves >= 2)      // (a tree of one leaf has no pairs: the walk starts at the leaf and the frame is never used)
                return fail(RT_ERR_STATE, "%s reads the packed pair table and this scene's hierarchy has none", inst.name);
            p.mat_in_lds = 0;
            lds = rt::lds_bytes_pairs(0, 0, false, n_samples, 1, 0, c->bvh.stack_depth, 64 * inst.waves) + 32;
            break;
        case rt::kTabPairsTopLds:           // header | the promoted top of the tree (n_top pairs = "n_top + 1 leaves") | stacks
            p.mat_in_lds = 0;
            lds = rt::lds_bytes_pairs(0, 0, false, n_samples, c->bvh.n_top + 1, 0, c->bvh.stack_depth, 64 * inst.waves);
            break;
        default:
            return fail(RT_ERR_STATE, "%s: unknown table kind %d", inst.name, inst.tables);
    }
#if RT_DIAGNOSTICS
    if (const char *pad = getenv("RT_LDS_PAD")) lds += (size_t)atoi(pad);       // occupancy experiments: fewer workgroups per CU
#endif
    if (lds > kLdsMax) return fail(RT_ERR_ARG, "%s needs %zu B of LDS for this scene (limit %zu)", inst.name, lds, kLdsMax);
    *lds_out = lds;
    return RT_OK;
}

// `form`: 0 = the context's choice, 1 = the hierarchy (if the scene has one), 2 = the plain sweep; 3 / 4 = the sweep WITH / WITHOUT cooperative
// any-hit whatever the sphere count says (the small-scene measurement below)
static int launch_form(rt_ctx *c, int n_samples, hipStream_t stream, int form, bool natural_order = false) {
    if (!c->have_scene || !c->have_cam)
        return fail(RT_ERR_STATE, "rt_set_scene and rt_set_camera must precede rendering");
    if (n_samples < 0) return fail(RT_ERR_ARG, "n_samples < 0");
    if (n_samples > 0x7fffffff - c->current_sample)
        return fail(RT_ERR_ARG, "pass counter would overflow (%d + %d)", c->current_sample, n_samples);
    if (n_samples == 0 || c->local_rows == 0) return RT_OK;
    int rc = chain(c, stream);
    if (rc != RT_OK) return rc;

    rt::LaunchParams p = make_params(c, n_samples);
    const size_t lds_all = rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, true, n_samples);
    // materials ride along in LDS only while that keeps at least 6 workgroups per CU resident
    // (160 KiB / 24 KiB); larger scenes read them from L2 once per hit
    p.mat_in_lds = lds_all <= (size_t)c->mat_lds_limit;
    const size_t lds_sweep = rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, p.mat_in_lds != 0, n_samples);

    // which instance: arithmetic mode x role x workgroup shape.  Single-wavefront workgroups (8x8 tiles) keep the wave
    // slots of a CU full (a 4-wavefront workgroup waits for four free slots at once) and give the heavy-first order a
    // finer granule; each stages its own copy of the tables, so only while 24 copies fit a CU.
    bool fast = c->mode == RT_MODE_FAST;
    const bool coop = form == 3 ? true : (form == 4 ? false : (c->coop_min > 0 && c->scene.n_spheres >= (uint32_t)c->coop_min));
    const bool w1 = c->wg_waves == 1 || (c->wg_waves == 0 && lds_sweep + (coop ? 1536u : 256u) <= 6 * 1024);   // + the instance's static LDS
    int role = coop ? rt::kRoleCoop : rt::kRolePlain, waves = w1 ? 1 : 4;
    if (form != 2 && form < 3 && bvh_usable(c)) {
        // large scenes: the walk over the hierarchy, from LDS while its tables leave room for five workgroups per CU
        role = bvh_fits_lds(c, n_samples) ? rt::kRolePairs : (bvh_pairs_fit_lds(c, n_samples) ? rt::kRolePairsMixed : rt::kRolePairsGlobal);
        waves = 4;
        if (c->regen_gate <= 0) p.regen_gate = c->walk_gate;
    } else if (!sweep_stages_tables(c, n_samples)) {
        // no hierarchy (or it lost the measurement) and a table beyond the sweep's LDS budget: the plain sweep over the table in HBM / L2
        role = rt::kRoleSweepGlobal;
        waves = 4;
    }
    p.walk_round = (c->walk_round & 0xff) | (c->walk_tail << 8);       // (one kernel argument: pair steps in a row | tail lanes << 8)
    const rt::Instance *inst = nullptr;
#if RT_DIAGNOSTICS
    if (c->persist != 0 && c->mode < 100) {
        role = coop ? rt::kRolePersistCoop : rt::kRolePersist;
        waves = 4;
    } else if (c->mode >= 100) {           // a row of the table by number (rt_set_mode checked the range)
        fast = c->mode >= 200;
        int n = 0;
        const rt::Instance *t = instances(fast, &n);
        inst = &t[c->mode - (fast ? 200 : 100)];
    }
#endif
    if (!inst) inst = find_role(fast, role, waves);
    // (the product library ships no 4-wavefront PLAIN sweep: below 12 spheres the tables always fit the single-wavefront budget above; should a
    // launch get here all the same, the cooperative instance of that shape renders any scene -- same bits, it only shares its shadow sweeps)
    if (!inst && role == rt::kRolePlain) inst = find_role(fast, rt::kRoleCoop, waves);
    if (!inst) return fail(RT_ERR_STATE, "this library holds no %s instance of role %d with %d wavefronts per workgroup", fast ? "fast" : "parity", role, waves);
    size_t lds_use = 0;
    rc = bind_tables(c, *inst, n_samples, p, &lds_use);
    if (rc != RT_OK) return rc;
    const bool persist = (inst->flags & rt::kInstPersistent) != 0;

    const int tile_w = 8 * inst->waves * ((inst->flags & rt::kInstTwoRays) ? 2 : 1);
    dim3 grid((unsigned)((c->w + tile_w - 1) / tile_w), (unsigned)((c->local_rows + rt::kTileH - 1) / rt::kTileH));
    // heavy tiles first: launches leave per-tile costs, and a launch of the same scene, camera and tile shape walks the tiles in descending order of
    // cost (sorted on the device, once per change of scene or camera).  A LONG launch (8 passes and more) replaces the costs with its own and sorts
    // from the last long launch's.  SHORT launches -- the reference's own regime is a pass per call, the adapter's display loop about a millisecond's
    // worth -- used to get nothing of this: no costs from fewer than 4 passes, no sort below 8.  Yet once the order exists it is worth as much to them
    // (complex.scn 12 %, C3 8-13 %, 8192 spheres 12-22 % on launches of 1 / 2 / 4 passes: profiles/r06_order_short_launches.jsonl).  So while the order is
    // missing or stale, short launches ADD their costs up in a window (launch flag bit 1: the kernel's epilogue adds instead of stores), and the short
    // launch that finds 16 passes' worth there sorts from them; with a valid order short launches do not touch the costs at all.
    constexpr int kLongLaunch = 8, kShortWindow = 16;
    const uint32_t n_tiles = grid.x * grid.y;
    const bool instance_logs_cost = (inst->flags & rt::kInstNoTileCost) == 0;
    const bool short_launch = n_samples < kLongLaunch;
    bool accumulate = false;
    if (c->use_order && c->d_tile_cost && n_tiles <= c->n_tiles && instance_logs_cost) {
        if (c->cost_tiles != n_tiles) {                                            // another tile shape: start over
            c->cost_valid = c->order_valid = false;
            c->cost_passes = 0;
        }
        const bool order_wanted = !c->order_valid || c->order_stale;
        // (a short launch sorts from a long launch's costs whenever the order is stale -- they still predict the next frame -- and from a window's once it is full)
        if (c->cost_valid && order_wanted && !natural_order && (!short_launch || !c->cost_window || c->cost_passes >= (uint32_t)kShortWindow)) {
            hipLaunchKernelGGL(rt_order_tiles_kernel, dim3(1), dim3(1024), 0, stream, c->d_tile_cost, c->d_order, n_tiles);
            HIP_TRY(hipGetLastError());
            c->order_valid = true;
            c->order_stale = false;
            if (short_launch && c->cost_window) c->cost_passes = 0;                 // (the window's costs are spent)
        }
        if (c->order_valid && !natural_order) p.order = c->d_order;
        if (!short_launch || !c->order_valid || c->order_stale) {                   // (a short launch under a valid order leaves the costs alone)
            p.tile_cost = c->d_tile_cost;
            accumulate = short_launch && c->cost_window && c->cost_passes > 0;
            if (accumulate) p.skip_pixels |= 2;
        }
    }
    if (persist) {
        // just enough workgroups to fill the machine; the tile queue (counters[30]) does the rest
        size_t per_cu = lds_use > 0 ? (160 * 1024) / (lds_use + 6 * 1024) : 6;
        if (per_cu > 6) per_cu = 6;
        if (per_cu < 1) per_cu = 1;
        size_t blocks = (size_t)c->n_cus * per_cu;
        const size_t needed = ((size_t)p.n_tiles + 3) / 4;
        if (blocks > needed) blocks = needed;
        grid = dim3((unsigned)blocks, 1, 1);
        HIP_TRY(hipMemsetAsync(c->d_counters + 30, 0, sizeof(unsigned long long), stream));
    }
#if RT_DIAGNOSTICS
    if (inst->role == rt::kRoleTimelog && c->d_timelog && c->timelog_used < c->timelog_cap) {
        p.timelog = c->d_timelog;
        p.seq = c->timelog_used++;
        p.tl_tag = c->timelog_tag;
        p.wavelog = ((size_t)grid.x * grid.y * 4 <= c->wavelog_cap) ? c->d_wavelog : nullptr;
    }
#endif
    const hipError_t e = rt::launch_instance(*inst, p, grid, lds_use, stream);
    if (e != hipSuccess)
        return fail(RT_ERR_HIP, "kernel launch failed: %s (%s, grid %ux%u, lds %zu B)", hipGetErrorString(e), inst->name, grid.x, grid.y, lds_use);
    c->current_sample += n_samples;
    c->launches += 1;
    c->scene_launches += 1;
    c->last_kernel = inst->name;
    c->last_coop = inst->role == rt::kRoleCoop || inst->role == rt::kRolePersistCoop;
    c->last_form = (inst->tables == rt::kTabPairsLds || inst->tables == rt::kTabPairsGlobal || inst->tables == rt::kTabPairsLdsSlotsGlobal ||
                    inst->tables == rt::kTabPairsTopLds || inst->tables == rt::kTabPairsPacked) ? 1 : 2;
    if (p.tile_cost) {
        if (!short_launch) {
            c->cost_window = false;
            c->cost_passes = (uint32_t)n_samples;
        } else if (accumulate) {
            c->cost_passes += (uint32_t)n_samples;
        } else {                                    // a short launch has replaced the costs with its own: a new window
            c->cost_window = true;
            c->cost_passes = (uint32_t)n_samples;
        }
        c->cost_valid = c->cost_passes >= 4;        // (fewer than four passes' worth orders nothing)
        c->cost_tiles = n_tiles;
    }
    c->seeds_default = false;           // this launch has written every seed pair the context renders
    c->pixels_current = c->pixel_write != 0;
    return RT_OK;
}

// Hierarchy or plain sweep for this scene?  The walk wins by 5x on a thousand spheres scattered over a plane and
// loses on a box packed with overlapping glass -- so it is measured, once per scene: four launches in the same
// (natural) tile order -- the hierarchy warm, the hierarchy timed, the sweep warm, the sweep timed, each timed one
// between two events -- and when both timings have arrived (asked without blocking) the form that took less time per
// pass renders the rest.  A blocking call with enough passes splits off 1 + 2 + 1 + 2 passes for the probes and waits
// for the verdict before it queues the rest (progressive passes equal one launch bit for bit).  The verdict is kept
// for the scene; device-resident updates keep it until the tree has changed size by a quarter or 256 updates have
// gone by (rearm_probe_if_changed).  In a multi-device context only the first shard measures; the others follow it.
constexpr int kProbeSteps = 4;          // hierarchy warm, hierarchy timed, sweep warm, sweep timed

void probe_poll(rt_ctx *c, bool wait) {
    if ((c->probing_coop ? c->coop_pick : c->bvh_pick) != 0 || c->probe_state < kProbeSteps) return;
    if (wait) {
        if (hipEventSynchronize(c->probe_ev[3]) != hipSuccess) return;
    } else if (hipEventQuery(c->probe_ev[3]) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    float a = 0.f, b = 0.f;
    if (hipEventElapsedTime(&a, c->probe_ev[0], c->probe_ev[1]) != hipSuccess || hipEventElapsedTime(&b, c->probe_ev[2], c->probe_ev[3]) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    const double ta = (double)a / c->probe_samples[0], tb = (double)b / c->probe_samples[1];
    if (c->probing_coop) {
        // arm 0 = cooperative any-hit, arm 1 = plain.  The plain instance -- what the threshold says -- keeps anything inside 4 %: the scenes the
        // sharing is for gain 7-14 %, the Demo scene loses 2-10 %, and a timed probe of a 1/8 shard of a 1080p frame lasts 65 microseconds (two of the
        // eight shards of profiles/r06_shard_prediction.jsonl's first run picked the slower form on a 2 % dead band)
        c->coop_pick = ta < 0.96 * tb ? 1 : 2;
        return;
    }
    c->probe_ms[0] = ta;
    c->probe_ms[1] = tb;
    c->bvh_pick = ta <= 1.05 * tb ? 1 : 2;      // (a dead band of 5 % towards the usual winner: no flipping on a tie)
    c->pick_estimated = false;
    c->probe_tree = c->bvh_n_tree;
    c->probe_always = c->bvh.n_always;
    c->probe_updates = 0;
}

static int launch_probe(rt_ctx *c, int n_samples, hipStream_t stream) {
    const int k = c->probe_state;               // 0, 1: hierarchy (warm, timed); 2, 3: plain sweep (warm, timed)
    const bool timed = (k & 1) != 0;
    const int arm = k >> 1;
    int rc = chain(c, stream);
    if (rc != RT_OK) return rc;
    if (timed) HIP_TRY(hipEventRecord(c->probe_ev[2 * arm], stream));
    rc = launch_form(c, n_samples, stream, c->probing_coop ? (arm == 0 ? 3 : 4) : (arm == 0 ? 1 : 2), true);
    if (rc != RT_OK) return rc;
    if (timed) {
        HIP_TRY(hipEventRecord(c->probe_ev[2 * arm + 1], stream));
        c->probe_samples[arm] = n_samples;
    }
    c->probe_state = k + 1;
    return RT_OK;
}

void rearm_probe(rt_ctx *c) {
    c->scene_frames = 0;
    c->scene_launches = 0;
    c->probe_acc = 0;
    c->coop_pick = 0;
    c->probing_coop = false;
    c->bvh_pick = 0;
    c->pick_estimated = false;
    c->probe_state = 0;
    c->probe_ms[0] = c->probe_ms[1] = 0.0;
    c->probe_updates = 0;
}

// after a device-resident update rebuilt the hierarchy: is the verdict still about this tree?
void rearm_probe_if_changed(rt_ctx *c) {
    if (c->probing_coop) return;                // (coop against plain on a scene without a hierarchy: an update cannot change the sphere count -- the verdict stays)
    if (c->bvh_pick == 0 && c->probe_state == 0) return;
    if (!c->bvh_ok) {
        rearm_probe(c);
        return;
    }
    const uint32_t tree = c->bvh_n_tree, always = c->bvh.n_always;        // spheres, not padded slots: the shaped tree of an upload has partial leaves
    auto moved = [](uint32_t now, uint32_t then) { return 4u * (now > then ? now - then : then - now) > then + 8u; };
    if (moved(tree, c->probe_tree) || moved(always, c->probe_always) || ++c->probe_updates >= 256) {
        rearm_probe(c);
        c->bvh_est_valid = false;               // (the areas were the uploaded tree's: the changed scene is measured)
    }
}

// A long launch that would walk its tiles in image order although their costs can be had -- the first frame of a scene --
// renders 4 of its passes first (they are passes of the frame like any other: progressive launches equal one launch bit for bit),
// which prices the tiles, and the rest heavy first.
// A renderer that draws one frame per scene would otherwise never leave image order ("Heavy tiles first").
constexpr int kPricePasses = 4, kPriceFrom = 24;
static int launch_priced(rt_ctx *c, int n_samples, hipStream_t stream, int form) {
    const bool explicit_mode = form == 0;
    bool priced = false;
    if (!explicit_mode && c->use_order && c->d_tile_cost && n_samples >= kPriceFrom && !c->order_valid && !c->cost_valid) {
        const int rc = launch_form(c, kPricePasses, stream, form);
        if (rc != RT_OK) return rc;
        n_samples -= kPricePasses;
        priced = true;
    }
    const int rc = launch_form(c, n_samples, stream, form);
    // The order this launch walked came from four passes' worth of costs; the launch itself has now left the costs of all its passes,
    // a better prediction of the next frame: the next long launch sorts once more from those (C2 2.63 -> 2.58 ms per steady frame,
    // the same on passes not seen before; profiles/r05_resort_after_pricing_ab.jsonl).
    if (rc == RT_OK && priced && c->order_valid) c->order_stale = true;
    return rc;
}

// The same question answered WITHOUT a launch, from the surface areas of the tree the host built at rt_set_scene (rt_bvh.hip):
// a random line through the root box is expected to visit  P = sum of area(inner node) / area(root)  pairs (and leaves in
// proportion), each ray sweeps the always-list besides, and the plain sweep tests all n spheres.  Predicted time per ray of
// the walk over that of the sweep, in units of one sphere test of the sweep:
//     ratio = (kEstPair * P + kEstAlways * n_always) / (n + kEstSweepFixed)
// The three weights are a least-squares fit (log ratio) to the probe's own timings of both forms on 32 scenes of four
// families -- spheres scattered on a plane, a closed box packed with mirror / glass spheres, a cloud in the air, the Demo
// scene plus scattered spheres; 64 to 1400 spheres -- tools/choice_calibration.py, profiles/r04k_choice_calibration.jsonl
// (this round's walk kernel): rms error 11 %, 9 % at worst between 0.55 and 1.8.  (A term for the expected leaf visits fitted to zero: they go with P.)
// Outside a band around 1 the estimate decides and nothing is measured -- a new scene's first frame then costs what a frame
// costs; inside it the four probe launches run as before.
constexpr double kEstPair = 18.7, kEstAlways = 8.9, kEstSweepFixed = 17.9;
constexpr double kEstBandLo = 0.75, kEstBandHi = 1.33;
double estimate_ratio(const rt_ctx *c) {
    return (kEstPair * c->bvh_est_pairs + kEstAlways * (double)c->bvh.n_always) / ((double)c->scene.n_spheres + kEstSweepFixed);
}

// Cooperative any-hit or not for a scene of fewer than coop_min spheres (rt_internal.h coop_pick): MEASURED, on the host's own launches as they
// come -- never split, never reordered, scheduled like any other (heavy tiles first): the sharing's worth depends on the order the tiles run in, so
// it is timed under the order the frames will run in.  Four steps: coop warm, coop timed, plain warm, plain timed; a step takes whole launches and
// ends once it holds enough passes -- a warm step one launch, a timed step 16 passes between its two events; a launch of 16 passes or more needs no
// warm step (it warms itself), so a host that renders whole frames spends its second frame on the cooperative instance, its third on the plain
// one, and has the verdict for the fourth; a host that queues a pass per call has it after 34 passes.  A scene's FIRST long blocking frame is not
// part of it (a new scene's first frame prices its tiles and runs partly in image order: it measures neither form's steady state), so a host that
// renders one frame per scene never runs the form the threshold would not have picked.  (The first version of this round split a long call into
// four short probe launches in natural tile order, as the hierarchy's probe does: +6 ... +8 % on that frame, and on shards of a frame, 65
// microseconds per timed probe, it picked the slower form on two shards of eight -- profiles/r06_coop_probe_first_frame.jsonl.)
constexpr uint32_t kCoopProbeFrom = 4;          // (below four spheres a shadow sweep has nothing to share out)
constexpr int kCoopTimedPasses = 16;
static int launch_small(rt_ctx *c, int n_samples, hipStream_t stream, bool may_block) {
    if (c->choice_leader)                       // a shard of a multi-device context: the form the first shard just launched
        return launch_priced(c, n_samples, stream, c->choice_leader->last_coop ? 3 : 4);
    const uint32_t n = c->scene.n_spheres;
    const bool open = c->coop_probe != 0 && c->coop_min > 0 && n >= kCoopProbeFrom && n < (uint32_t)c->coop_min && c->wg_waves == 0 && c->persist == 0 &&
                      tables_fit_lds(c, n_samples);
    if (!open) return launch_priced(c, n_samples, stream, 2);
    c->probing_coop = true;
    probe_poll(c, false);
    if (c->coop_pick != 0) return launch_priced(c, n_samples, stream, c->coop_pick == 1 ? 3 : 4);
    if (c->probe_state == kProbeSteps) return launch_priced(c, n_samples, stream, 4);   // both timings queued, not back yet: what the threshold says meanwhile
    const bool whole_frame = may_block && n_samples >= kCoopTimedPasses;
    if (whole_frame && c->scene_frames == 0 && c->probe_state == 0) return launch_priced(c, n_samples, stream, 2);
    if (whole_frame && (c->probe_state & 1) == 0) c->probe_state += 1;                  // (a long launch warms itself)
    const int k = c->probe_state, arm = k >> 1;
    const bool timed = (k & 1) != 0;
    int rc = chain(c, stream);
    if (rc != RT_OK) return rc;
    if (timed && c->probe_acc == 0) HIP_TRY(hipEventRecord(c->probe_ev[2 * arm], stream));
    rc = launch_priced(c, n_samples, stream, arm == 0 ? 3 : 4);
    if (rc != RT_OK) return rc;
    if (!timed) {
        c->probe_state = k + 1;
        return RT_OK;
    }
    c->probe_acc += n_samples;
    if (c->probe_acc >= kCoopTimedPasses) {
        HIP_TRY(hipEventRecord(c->probe_ev[2 * arm + 1], stream));
        c->probe_samples[arm] = c->probe_acc;
        c->probe_acc = 0;
        c->probe_state = k + 1;
    }
    return RT_OK;
}

int launch(rt_ctx *c, int n_samples, hipStream_t stream, bool may_block) {
    if (c->tables_stale) {              // records updated on the device since the tables were built: build them now, once
        const int rc = refresh_tables(c, stream);
        if (rc != RT_OK) return rc;
    }
    const bool measured = c->walk_forced == 0 && c->mode < 100;
    if (measured && n_samples > 0 && c->local_rows != 0 && c->have_scene && c->have_cam && !bvh_usable(c)) return launch_small(c, n_samples, stream, may_block);
    if (!measured || n_samples <= 0 || c->local_rows == 0 || !c->have_scene || !c->have_cam || !bvh_usable(c))
        return measured ? launch_priced(c, n_samples, stream, 2) : launch_form(c, n_samples, stream, 0);
    // no probe where the answer is known and asking is dear: from 1500 spheres in the tree on the hierarchy won on every
    // scene measured, open or packed, and one pass of the sweep at 1080p costs 2.6 ms at 1024 spheres,
    // 28 ms at 4096, 138 ms at 8192 from staged tables (through the scalar cache, round 6: a fifth of that at 8192 -- still tens of milliseconds a pass)
    if (c->bvh_n_tree >= kAlwaysWalkFrom || !tables_fit_lds(c, n_samples)) return launch_priced(c, n_samples, stream, 1);
    if (c->choice_leader)                       // a shard of a multi-device context: the form the first shard just launched
        return launch_priced(c, n_samples, stream, c->choice_leader->last_form == 2 ? 2 : 1);
    probe_poll(c, false);
    if (c->bvh_pick == 0 && c->probe_state == 0 && c->use_estimate && c->bvh_est_valid) {
        const double r = estimate_ratio(c);
        c->est_ratio = r;
        if (r < kEstBandLo || r > kEstBandHi) {
            c->bvh_pick = r < 1.0 ? 1 : 2;
            c->pick_estimated = true;
            c->probe_tree = c->bvh_n_tree;
            c->probe_always = c->bvh.n_always;
            c->probe_updates = 0;
        }
    }
    if (c->bvh_pick != 0) return launch_priced(c, n_samples, stream, c->bvh_pick);
    if (c->probe_state == kProbeSteps) return launch_form(c, n_samples, stream, 1);    // probes in flight: the usual winner meanwhile
    if (may_block && n_samples >= 16) {
        int done = 0;
        while (c->probe_state < kProbeSteps) {
            const int k = (c->probe_state & 1) ? 2 : 1;
            const int rc = launch_probe(c, k, stream);
            if (rc != RT_OK) return rc;
            done += k;
        }
        probe_poll(c, true);
        return launch_priced(c, n_samples - done, stream, c->bvh_pick ? c->bvh_pick : 1);
    }
    return launch_probe(c, n_samples, stream);
}

// a shard's launch on its own stream, for the multi-device context (rt_multi.hip)
int render_shard(rt_ctx *c, int n_samples, bool may_block) {
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    return launch(c, n_samples, c->stream, may_block);
}

}  // namespace rt
