// rt_internal.h -- what the translation units of the library share behind the C ABI: the context
// record, error reporting, and the hooks of the multi-device context (rt_multi.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

#include "rt_device.h"

struct rt_multi;                    // rt_multi.hip

struct rt_ctx {
    int device = 0;
    int w = 0, h = 0;
    int rank = 0, nranks = 1, tile_rows = 8, local_rows = 0;
    rt_multi *multi = nullptr;          // non-null: this record is the front of a multi-device context
    uint32_t *d_seeds = nullptr;
    uint32_t *d_seeds0 = nullptr;       // pristine default stream, for device-side resets
    float *d_colors = nullptr;
    uint32_t *d_pixels = nullptr;
    uint32_t *d_pixels_ext = nullptr;   // caller-owned target of rt_set_pixel_buffer, or null
    void *pinned_out = nullptr;         // host buffer page-locked by rt_pin_output, or null
    int pixel_write = 1;                // rt_set_pixel_write
    bool pixels_current = true;         // the packed pixel buffer holds the frame of the running average
    bool seeds_default = false;         // after rt_reset_async: the next launch reads the pristine stream
    unsigned long long *d_counters = nullptr;
    unsigned long long *d_stats = nullptr;      // rt::kStatReplicas x 8 partial work counters
    // scene: raw records + the tables the device-side build kernel derives from them, one allocation
    rt_sphere *d_spheres = nullptr;
    float4 *d_tables = nullptr;         // geom | emis | colr | lightA | lightB, each `scene_cap` entries
    uint32_t scene_cap = 0;
    // hierarchy over the small spheres of a large scene (rt_device.h BvhTables), rebuilt with the tables
    float4 *d_bvh = nullptr;            // blob, sized for scene_cap
    rt::BvhTables bvh{};
    bool bvh_ok = false;                // the blob describes the current scene
    float4 *h_bvh_stage = nullptr;      // page-locked buffer of the host-side build (trees beyond the one-workgroup device build)
    size_t bvh_stage_cap = 0;           // float4
    hipEvent_t bvh_stage_ev = nullptr;
    bool bvh_stage_used = false;
    // records that repeat an EARLIER record bit for bit in what a ray test reads ({centre, radius^2}) stay out of the hierarchy (rt_bvh.hip mark_duplicates):
    // one byte per record on the device, uploaded only while the scene has such records
    uint8_t *d_dup = nullptr, *h_dup_stage = nullptr;       // [scene_cap] each; the second page-locked
    hipEvent_t dup_ev = nullptr;
    bool dup_stage_used = false, have_dups = false;
    uint32_t n_dups = 0;
    int bvh_sah = 1;                    // the hierarchy's shape is chosen by surface area (rt_bvh.hip): 1 = uploads below kAlwaysWalkFrom tree spheres on the host,
                                        // larger uploads and every device-resident update on the device; 2 = the device for uploads too; 0 = the fixed (halved) shape
    uint32_t bvh_n_tree = 0;            // spheres inside the tree (the slots are padded to whole leaves)
    // surface-area sums of a host-built tree (rt_bvh.hip): what a random line through the root box is expected to visit --
    // pair steps (inner nodes, the root counted once) and leaves; the estimate that settles hierarchy against sweep without a launch
    double bvh_est_pairs = 0.0, bvh_est_leaves = 0.0;
    bool bvh_est_valid = false;
    int bvh_min = 56;                   // scenes with at least this many spheres inside the tree use it (0 = never)
    int bvh_lds_limit = 31 * 1024;      // its tables are staged in LDS while five workgroups of that size fit a CU (2048 spheres: 6.2 ms from L2 with
                                        // 4-5 waves per SIMD against 9.5 ms from LDS with two workgroups per CU); larger ones are read from HBM / L2
    int bvh_top_pairs = (int)rt::kBvhTopPairs;   // pairs promoted to the front of the table for the walk that reads it from HBM / L2 (rt_bvh.hip promote_top; diagnostics knob: 0 = none)
    int bvh_packed = 1;                 // the packed pair table is built for trees the walk reads from HBM / L2 (rt_bvh.hip pack_pairs; diagnostics knob: 0 = never)
    int bvh_mixed = 1;                  // tables beyond that limit whose PAIRS fit it: pairs staged, slots from HBM / L2 (rt_trace_*_pairs_m); diagnostics knob: 0 = everything from L2
    int walk_gate = 16, walk_round = 4; // rt_walk.inc.h: ready lanes that make the wavefront shade; pair steps in a row before a leaf step
                                        // (round 4, this kernel: 2 / 3 / 4 / 6 in a row = 5.42 / 5.37 / 5.22 / 5.45 ms on C3, profiles/r04k_walk_sweep.jsonl)
    int walk_tail = 0;                  // lanes that may be left walking when a trip's walk phase ends (0 = none: every walk runs to its end within the trip)
    int walk_forced = 0;                // 0 = measured choice (below); diagnostics: 1 = the hierarchy whenever the scene has one
    // hierarchy or plain sweep?  Decided per scene by measurement (rt_api.hip launch()): each form once warm and once
    // timed between events, in the same tile order; whichever took less time per pass renders the rest
    int bvh_pick = 0;                   // 0 = not decided yet, 1 = hierarchy, 2 = plain sweep
    bool pick_estimated = false;        // ... and it came from the surface-area estimate, not from a measurement
    int use_estimate = 1;               // diagnostics knob: 0 = every undecided scene is measured
    double est_ratio = 0.0;             // the estimate's predicted walk / sweep time (0 = none made)
    int probe_state = 0;                // probe launches issued (0..4)
    // ... and for a scene WITHOUT a hierarchy and fewer spheres than coop_min: cooperative any-hit or not, timed on the host's own launches (rt_launch.hip launch_small).
    // Below 12 spheres the threshold alone picks wrongly either way -- the Demo scene is 2 % faster without the sharing, the reference's simple.scn,
    // caustic.scn and caustic3.scn (6 and 10 records) 7-11 % faster with it (profiles/r06_reference_scenes.jsonl) -- so it is measured.
    int coop_pick = 0;                  // 0 = not decided yet, 1 = cooperative any-hit, 2 = plain
    bool probing_coop = false;          // the probe launches in flight time coop against plain (not hierarchy against sweep)
    bool last_coop = false;             // the last launch was a cooperative any-hit instance (what a multi-device context's other shards follow)
    uint32_t scene_frames = 0;          // resets since rt_set_scene that followed at least one launch OF THAT SCENE: frames of it already rendered
    uint64_t scene_launches = 0;        // launches since rt_set_scene
    int probe_acc = 0;                  // passes launched so far inside the timed step of the coop / plain measurement that is open
    int coop_probe = 1;                 // diagnostics knob: 0 = the threshold alone decides (round 5's behaviour)
    uint32_t probe_tree = 0, probe_always = 0;   // the tree the verdict was measured on
    int probe_updates = 0;              // device-resident updates since the verdict (it is measured again after 256)
    rt_ctx *choice_leader = nullptr;    // a shard of a multi-device context: the shard whose verdict it follows (null: its own)
    int probe_samples[2] = { 0, 0 };
    double probe_ms[2] = { 0.0, 0.0 };  // measured time per pass: hierarchy, plain sweep (0 = not measured)
    hipEvent_t probe_ev[4] = { nullptr, nullptr, nullptr, nullptr };
    std::vector<unsigned char> is_light;   // host mirror of the light test per sphere (sizes the light list)
    std::vector<rt_sphere> h_spheres;      // host mirror of the records (an identical rt_set_scene uploads nothing)
    // heavy-first tile order (rt_trace.inc.h): per-tile cost of the last launch, and the order derived from it
    uint32_t *d_tile_cost = nullptr, *d_order = nullptr;
    uint32_t n_tiles = 0;               // capacity of the two arrays (8x8 tiles)
    uint32_t cost_tiles = 0;            // tile count of the launch the costs come from
    int wg_waves = 0;                   // diagnostics knob: 0 = automatic, 1 / 4 = force the workgroup shape
    bool cost_valid = false, order_valid = false;
    uint32_t cost_passes = 0;           // passes behind the costs now in d_tile_cost
    bool cost_window = false;           // ... which come from a window of SHORT launches (fewer than 8 passes each) adding up, not from one long launch
    bool order_stale = false;           // scene or camera have changed since the order was sorted: it stays in use until a long launch sorts it again
    int use_order = 1;
    rt_sphere *h_stage = nullptr;       // page-locked staging ring for sphere uploads
    uint32_t stage_cap = 0;             // records per slot
    int stage_next = 0;
    hipEvent_t stage_ev[4] = { nullptr, nullptr, nullptr, nullptr };
    bool stage_used[4] = { false, false, false, false };
    rt::SceneTables scene{};
    rt_camera cam{};
    bool have_scene = false, have_cam = false;
    int mode = RT_MODE_PARITY;
    int regen_gate = 0;                 // 0 = choose from the scene size
    int mat_lds_limit = 24 * 1024;
    bool tables_stale = false;          // rt_update_spheres_async has changed records since the tables and the hierarchy were built: refresh_tables() builds them
                                        // ONCE, on the stream of whatever reads them next, however many updates went by (200 moved spheres by 200 calls cost 200
                                        // rebuilds before: 353 ms a frame at 8192 spheres against 4.9 in one call -- profiles/r06_update_calls.jsonl)
    int sweep_lds_limit = 40 * 1024;    // the plain / cooperative sweep stages its tables while four workgroups of that size fit a CU; beyond, the table is read through the
                                        // scalar cache at six wavefronts per SIMD whatever its size (rt_trace_*_g).  Measured at 1080p on scenes without a hierarchy
                                        // (profiles/r06_g_threshold.jsonl): at 48 KB (3 per CU) rt_trace_*_g takes 0.62 / 0.92 x the staged sweep's time (NaN records / a closed
                                        // box of mirrors), at 64 KB (2) 0.41 / 0.63, at 96 KB and more (1) 0.19; at 32 KB (4) 0.80 / 1.12, below that 0.9 ... 1.3
    int coop_kmax = 0;                  // cooperative any-hit only while no more than this many shadow rays are pending in the wavefront (0 = no limit)
    int coop_min = 12;                  // scenes with at least this many spheres use the cooperative any-hit instance (0 = never)
    int persist = 0;                    // diagnostics: persistent-wavefront instances
    int n_cus = 256;
    int current_sample = 0;
    uint64_t launches = 0;
    const char *last_kernel = "";       // symbol of the instance the last launch used
    int last_form = 0;                  // ... 1 = it walked the hierarchy, 2 = a plain sweep (0 = nothing launched yet)
    double last_ms = 0.0;
    unsigned long long debug_counters[24] = {};   // diagnostic instances only
    hipStream_t stream = nullptr;       // the context's own (non-blocking) stream
    hipStream_t last_stream = nullptr;  // stream of the most recent launch / update (what readers wait for)
    bool used_foreign_stream = false;   // some launch went to a caller's stream
    bool abandon_streams = false;       // a shard of a multi-device context whose gather failed: its stream may hold a transfer that never
                                        // completes, so rt_destroy does not wait for it
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_dep = nullptr;
    // rt_throttle: the rt_render_async launches in flight, each between two events (a ring, oldest at flight_next)
    struct Flight {
        hipEvent_t start = nullptr, stop = nullptr;
        int n_samples = 0;
        bool pending = false;
    };
    static constexpr int kFlights = 8;
    Flight flight[kFlights];
    int flight_next = 0;
    bool throttle_on = false;
    double flight_ms_per_pass = 0.0;    // device time per pass of the most recent finished launch
    // diagnostics build: device wall-clock logs
    unsigned long long *d_timelog = nullptr, *d_wavelog = nullptr, *d_blocklog = nullptr;
    uint32_t *d_stalelog = nullptr;
    uint32_t timelog_cap = 0, timelog_used = 0, wavelog_cap = 0;
    unsigned long long timelog_tag = 0;
};

namespace rt {

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return rt::fail(RT_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                               \
    } while (0)

// ---- rt_api.hip: the context's device and the order of its work ----
int select_device(const rt_ctx *c);
int chain(rt_ctx *c, hipStream_t stream);          // `stream` waits for whatever the context queued last elsewhere (ALL its work runs in issue order)
int wait_all(rt_ctx *c);                            // host waits for everything the context has queued
const double *create_breakdown();                   // host ms of the last rt_create by phase (rt_debug_create_breakdown)

// ---- rt_launch.hip: one launch of the render kernel ----
constexpr uint32_t kAlwaysWalkFrom = 1500;          // tree spheres from which the hierarchy is walked without estimate or measurement (rt_launch.hip)
constexpr size_t kLdsMax = 152 * 1024;              // what the kernels' dynamic-LDS attribute allows
LaunchParams make_params(rt_ctx *c, int n_samples);
const Instance *instances(bool fast, int *count);
bool tables_fit_lds(const rt_ctx *c, int n_samples);
void probe_poll(rt_ctx *c, bool wait);              // hierarchy against sweep: the measurement's verdict, if its events have completed
void rearm_probe(rt_ctx *c);                        // a new scene: undecided again
void rearm_probe_if_changed(rt_ctx *c);             // after a device-resident update rebuilt the hierarchy
int refresh_tables(rt_ctx *c, hipStream_t stream);     // rt_scene.hip: tables and hierarchy from the records as they are now, if updates made them stale
double estimate_ratio(const rt_ctx *c);             // predicted walk / sweep time per ray from the uploaded tree's surface areas
int launch(rt_ctx *c, int n_samples, hipStream_t stream, bool may_block = false);

// ---- rt_scene.hip: records, tables, staging ----
void free_scene(rt_ctx *c);
int ensure_scene_capacity(rt_ctx *c, uint32_t count);
int upload_spheres(rt_ctx *c, uint32_t first, uint32_t count, const rt_sphere *spheres, uint32_t n_total, hipStream_t stream, bool full_upload);

// the hierarchy of large scenes (rt_bvh.hip): per-device set-up of the build kernel; build on `stream` for the scene the
// context's host mirror holds (sets c->bvh / c->bvh_ok; nothing is read back)
hipError_t prepare_bvh_build();
int build_bvh(rt_ctx *c, uint32_t n_total, hipStream_t stream, bool full_upload = false);
int render_shard(rt_ctx *c, int n_samples, bool may_block);      // rt_api.hip: one shard's launch on its own stream

// multi-device context (rt_multi.hip); `front` is the rt_ctx whose `multi` points at the record
void multi_destroy(rt_ctx *front);
int multi_set_scene(rt_ctx *front, const rt_sphere *spheres, uint32_t count);
int multi_update_spheres(rt_ctx *front, uint32_t first, uint32_t count, const rt_sphere *spheres);
int multi_set_camera(rt_ctx *front, const rt_camera *cam);
int multi_set_mode(rt_ctx *front, int mode);
int multi_reset(rt_ctx *front, bool async);
int multi_render(rt_ctx *front, uint32_t *out_host, int n_samples, bool blocking);
int multi_set_pixel_write(rt_ctx *front, int enable);
int multi_read_pixels(rt_ctx *front, uint32_t *out_host);
int multi_read_colors(rt_ctx *front, float *out_host);
int multi_read_seeds(rt_ctx *front, uint32_t *out_host);
int multi_get_stats(rt_ctx *front, rt_stats *out);
int multi_device_pixels(rt_ctx *front, void **dptr, size_t *count);
int multi_pin_output(rt_ctx *front, uint32_t *out_host, size_t count);
void *multi_stream(rt_ctx *front);
int multi_shards(const rt_ctx *front);
int multi_wait_frame(rt_ctx *front);
const char *multi_last_kernel(const rt_ctx *front);
rt_ctx *multi_first_shard(rt_ctx *front);
rt_ctx *multi_shard(rt_ctx *front, int r);
int multi_debug_each(rt_ctx *front, int (*fn)(rt_ctx *, int), int arg);
int multi_debug_break(rt_ctx *front);
int multi_debug_set_rccl(const char *path, int repeated_counts_as_distinct);   // diagnostics build

}  // namespace rt
