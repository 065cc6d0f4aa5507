// rt_device.h -- launch parameters shared by the host shim (rt_api.hip) and the two kernel
// translation units (parity: no contraction; fast: contraction + hardware transcendentals).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_api.h"

namespace rt {

constexpr int kBlockThreads = 256;  // 4 wavefronts
constexpr int kTileW = 32;          // block tile: 32 x 8 pixels, one 8x8 sub-tile per wavefront
constexpr int kTileH = 8;
constexpr int kMaxDepth = 8;        // .cl:320 (depth > 7 ends the path)
constexpr int kStatReplicas = 64;    // work counters are summed into 64 separate 64-byte lines
constexpr int kMaxK2Table = 1024;   // running-average reciprocals kept in LDS up to this many passes per launch

// Sphere tables in HBM, written once by rt_set_scene and staged into LDS by every workgroup.
//   geom[i] = { p.x, p.y, p.z, rad*rad }      closest-hit / any-hit loops read only this
//   emis[i] = { e.x, e.y, e.z, bits(refl) }   read once per hit, for the sphere that was hit
//   colr[i] = { c.x, c.y, c.z, rad }
//   lightA[j] = { p.x, p.y, p.z, rad }         the j-th sphere, in scene order, whose emission
//   lightB[j] = { e.x, e.y, e.z, 4*pi*rad*rad } passes the reference's zero test (.cl:135-138
//                                              looks at x and z only): the light list that
//                                              SampleLights (.cl:249-303) walks
struct SceneTables {
    const float4 *geom;
    const float4 *emis;
    const float4 *colr;
    const float4 *lightA;
    const float4 *lightB;
    uint32_t n_spheres;
    uint32_t n_lights;
};

// Bounding-volume hierarchy over the small spheres of a large scene (built by rt_bvh.hip, walked by rt_walk.inc.h).
// It only decides WHICH spheres a ray is tested against: every test that is made is the reference's arithmetic, and
// the selection rule (smallest distance, lowest scene index among equals; lowest blocking index for shadow rays) is
// the reference's loop order restated, so frames and counters do not change.  One blob in HBM:
//   hdr[0] = { root box centre, half diagonal }   hdr[1] = { min |rad|, max |rad|, 1 / (2 min |rad|), bits(root pair | complaints << 16) }
//   slots[j] = { p, rad*rad } for j < n_always: the spheres that stay outside the tree (large or non-finite), in
//            scene order, swept by every ray as before; then kBvhLeaf per leaf, in leaf order (padded with NaN records)
//   index[j] = scene index of slot j (u32; read from HBM / L2 for accepted candidates only)
//   pairs[4m .. 4m+3] = the two children of inner node m as { lo.xyz, bits(ref) }, { hi.xyz, bits(lowest scene index) }
//            twice; ref = kBvhLeafRef | leaf number, or the number of the child's own pair.  The walk takes the nearer
//            child first and keeps the other on a per-lane stack.  Inner node m - 1 is the one that splits its range
//            of leaves in front of leaf m, wherever that split lies: the device build halves every range (root = pair
//            n_leaves / 2 - 1), the host build of a full scene upload cuts by surface area (rt_bvh.hip); BvhTables::root says
//            which pair the walk starts at -- it travels in the header (hdr[1].w), because a tree shaped on the device has it where the host cannot see it (a tree of one leaf has no pairs: kBvhLeafRef).
//   emis[j], colr[j] = the material records of slot j ({ emission, bits(refl) }, { colour, radius }: SceneTables' records in SLOT
//            order): a closest hit reads its material by the slot the walk ended on -- one round trip to L2 instead of two
//            (scene index first, then the record by index)
#ifndef RT_BVH_LEAF
#define RT_BVH_LEAF 8                   /* spheres per leaf (4 measured in round 4: tools/leaf_size_ab.sh) */
#endif
constexpr int kBvhLeaf = RT_BVH_LEAF;
constexpr uint32_t kBvhLeafRef = 0x8000u;
struct BvhTables {
    const float4 *blob;     // hdr | slots | index | pairs | emis by slot | colr by slot
    uint32_t n_always, n_leaves, n_slots;
    uint32_t stack_depth;   // entries a lane's stack needs (tree depth + 1; an upper bound for trees shaped on the device)
    uint32_t emis_at;       // bvh_emis_at(n_leaves, n_slots), formed on the host: the kernel has no scalar registers to spare for it
    uint32_t n_top;         // pairs [0, n_top) are the TOP of the tree in breadth-first order (pair 0 = the root's), the rest keep the builders' order
                            // behind them (rt_bvh.hip rt_bvh_promote_kernel): the walk that reads its tables from HBM / L2 stages exactly these in LDS.
                            // 0 = the builders' numbering, nothing promoted
    uint32_t packed_at;     // float4 offset into the blob of the PACKED pair table (rt_bvh.hip rt_bvh_pack_pairs_kernel), 0 = none: frame { r0.xyz, - },
                            // { scale.xyz, - }, then 32 bytes per pair in the pairs' own order -- twelve 16-bit grid coordinates of the two child
                            // boxes (plane = r0 + q * scale, lows rounded down and highs up by a whole cell), the two references, the two lowest
                            // scene indices >> kBvhLowShift.  Half the bytes per pair step for the walk that reads its tables from HBM / L2
};
constexpr uint32_t kBvhLowShift = 2;        // (262 144 scene indices in 16 bits: a lower bound of the lowest index below a child prunes as safely as the index itself)
constexpr uint32_t kBvhTopPairs = 255;      // eight full levels: 16 KiB of LDS beside the stacks, five workgroups per CU still fit
// offsets into the blob, in float4 units
__host__ __device__ inline uint32_t bvh_slots_at() { return 2u; }
__host__ __device__ inline uint32_t bvh_index_at(uint32_t n_slots) { return 2u + n_slots; }
__host__ __device__ inline uint32_t bvh_pairs_at(uint32_t n_slots) { return 2u + n_slots + (n_slots + 3u) / 4u; }
__host__ __device__ inline uint32_t bvh_emis_at(uint32_t n_leaves, uint32_t n_slots) { return bvh_pairs_at(n_slots) + 4u * (n_leaves ? n_leaves - 1u : 0u); }
__host__ __device__ inline uint32_t bvh_colr_at(uint32_t n_leaves, uint32_t n_slots) { return bvh_emis_at(n_leaves, n_slots) + n_slots; }
inline size_t bvh_blob_float4s(uint32_t n_leaves, uint32_t n_slots) { return (size_t)bvh_colr_at(n_leaves, n_slots) + (size_t)n_slots; }
// LDS of the instance that walks the pairs: hdr | pairs | slots | per-lane stacks (u16) for `threads` lanes
inline size_t lds_bytes_pairs(uint32_t n_spheres, uint32_t n_lights, bool mat_in_lds, int n_samples, uint32_t n_leaves,
                              uint32_t n_slots, uint32_t stack_depth, int threads) {
    size_t b = (2 + 4 * (size_t)(n_leaves ? n_leaves - 1 : 0) + n_slots) * 16 + (size_t)n_lights * 32;
    b += (((size_t)stack_depth * threads * 2) + 15) & ~(size_t)15;
    if (mat_in_lds) b += (size_t)n_spheres * 32;
    if (n_samples <= kMaxK2Table) b += (size_t)(n_samples > 0 ? n_samples : 0) * 4;
    return (b + 15) & ~(size_t)15;
}

struct LaunchParams {
    SceneTables scene;
    rt_camera cam;
    uint32_t *seeds;        // [2*w*h], pair per pixel at gid = y*w + x        (.cl:570-571); written at the end of a launch
    const uint32_t *seeds_in;   // read at the start of a launch: `seeds`, or the pristine default stream for the
                                // first launch after rt_reset_async (no copy, and no dependence on one having landed)
    float *colors;          // [3*w*h], running average at (h-1-y)*w + x       (.cl:579)
    uint32_t *pixels;       // [local_rows*w], packed RGBX of this rank's rows (.cl:594)
    unsigned long long *stats;     // [kStatReplicas][8] u64: samples, closest, shadow, tests, draws (per-replica partial sums)
    unsigned long long *counters;  // 32 u64: diagnostics [8..29], tile queue head [30]
    int w, h;
    int first_sample, n_samples;
    int rank, nranks, tile_rows, local_rows;
    int mat_in_lds;         // material tables staged into LDS as well (fits 64 KiB)
    int n_tiles, tiles_x;   // persistent instances: 8x8 pixel tiles of this rank's rows, and tiles per row
    int regen_gate;         // lanes that must be waiting before finished lanes start new paths (1 = free-running)
    int coop_kmax;          // cooperative any-hit: with more pending shadow rays than this in the wavefront the lanes sweep for themselves (0 = no limit)
    float inv_w, inv_h;     // 1.f / w, 1.f / h (.cl:503-504), divided once on the host: kernel arguments live in SGPRs
    int skip_pixels;        // launch flags.  bit 0: this launch leaves the packed pixels alone (rt_set_pixel_write(ctx, 0)); bit 1: it ADDS its per-tile costs
                            // to the ones in `tile_cost` instead of replacing them (a short launch inside an accumulation window, rt_launch.hip)
    const uint32_t *order;  // heavy-first walk of the 32x8 tiles (tile id = by * gridDim.x + bx), or null = natural order
    uint32_t *tile_cost;    // per tile: wall-clock ticks (10 ns) of its slowest wavefront, written by every launch (or null)
    // diagnostics build only (null in the product library): launch sequence number and the buffers the
    // instrumented instance logs device wall-clock intervals into (tools/gather_stress.py)
    unsigned long long *timelog;   // [seq][8]: min start, max end of the launch (s_memrealtime, 100 MHz), kind, tag, ...
    unsigned long long tl_tag;
    unsigned long long *wavelog;   // [workgroup*4 + wave][3]: start, end, xcc_id << 32 | HW_ID
    uint32_t seq;
    // instances that walk the hierarchy of a large scene (RT_OPT_BVH) -- at the end: the other instances' argument
    // offsets, and with them their scalar loads and SGPR allocation, are what they were without it
    int walk_round;         // rt_walk.inc.h: pair steps in a row before the leaf step of the lanes that hold a leaf
    BvhTables bvh;
};

// LDS bytes the kernels need for a scene
inline size_t lds_bytes(uint32_t n_spheres, uint32_t n_lights, bool mat_in_lds, int n_samples = 0) {
    size_t b = (size_t)n_spheres * 16 + (size_t)n_lights * 32;
    if (mat_in_lds) b += (size_t)n_spheres * 32;
    if (n_samples <= kMaxK2Table) b += (size_t)(n_samples > 0 ? n_samples : 0) * 4;
    return (b + 15) & ~(size_t)15;
}

// Final-frame pack kernels (rt_read_pixels): pixels[lrow*w + x] = toInt of the colour plane, with the
// arithmetic of the mode that renders (.cl:34,594-596), for frames whose launches skipped the pixel store.
hipError_t launch_pack_parity(const LaunchParams &p, hipStream_t stream);
hipError_t launch_pack_fast(const LaunchParams &p, hipStream_t stream);

// ---- the kernel instances ------------------------------------------------------------------------------
// One row per instance, written next to the instantiations in rt_kernel_parity.hip / rt_kernel_fast.hip: what the
// instance is and what it needs.  rt_api.hip picks an instance by ROLE, sizes its LDS from `tables` and refuses a
// launch whose instance needs a table the context does not have -- nothing in the host code depends on the order of
// the rows.  The product library holds the shipped rows only; the diagnostics build (librt_hip_diag.so) adds the
// A/B and verification instances (rt_set_mode 100 + row / 200 + row, or by name: rt_debug_instance).
enum InstanceTables : uint8_t {
    kTabSweepLds = 0,       // geometry + light tables staged in LDS (lds_bytes)
    kTabSweepGlobal = 1,    // ... read where they lie in HBM / L2
    kTabPairsLds = 2,       // the hierarchy (pairs, slots, stacks) staged in LDS (lds_bytes_pairs); needs BvhTables
    kTabPairsGlobal = 3,    // ... pairs and slots read where they lie; staged: header and stacks; needs BvhTables
    kTabPairsLdsSlotsGlobal = 4,   // ... the pairs staged, the slots read where they lie (the pairs fit the LDS budget, the whole tables do not)
    kTabPairsTopLds = 5,    // ... pairs and slots read where they lie but for the promoted top of the tree (BvhTables::n_top pairs), which is staged
    kTabPairsPacked = 6,    // ... the PACKED pair table (BvhTables::packed_at) and the slots read where they lie; staged: header, frame and stacks
};
enum InstanceRole : uint8_t {
    kRoleNone = 0,          // diagnostics: reachable by row / name only
    kRolePlain,             // small scenes
    kRoleCoop,              // 12 spheres and more: cooperative any-hit
    kRolePairs,             // large scenes through the hierarchy
    kRolePairsGlobal,       // ... tables beyond the LDS budget
    kRolePairsMixed,        // ... whose pairs still fit it (rt_trace_*_pairs_m)
    kRoleSweepGlobal,       // no hierarchy (or it lost the measurement) and a table beyond the sweep's LDS budget (rt_internal.h sweep_lds_limit)
    kRolePersist,           // diagnostics: persistent wavefronts (rt_debug_set_persist)
    kRolePersistCoop,
    kRoleTimelog,           // diagnostics: the shipped shape + device wall-clock logging
};
enum InstanceFlags : uint8_t {
    kInstPersistent = 1,    // the grid only fills the machine; tiles come from the queue at counters[30]
    kInstNoTileCost = 2,    // neither reads the heavy-first order nor leaves per-tile costs
    kInstStaticCoop = 4,    // carries the cooperative any-hit mailbox (1.5 KiB of static LDS per wavefront)
    kInstTwoRays = 8,       // diagnostics: two pixels per lane -- a wavefront's tile is 16 x 8, and every lane has two stacks
};
struct Instance {
    void (*fn)(const LaunchParams);
    const char *name;       // the kernel's symbol: what rocprofv3 lists and rt_last_kernel returns
    uint8_t waves;          // wavefronts per workgroup: 4 (32x8 tile) or 1 (8x8 tile)
    uint8_t tables;         // InstanceTables
    uint8_t role;           // InstanceRole (with `waves`, what launch() selects by)
    uint8_t flags;          // InstanceFlags
};
const Instance *parity_instances(int *count);
const Instance *fast_instances(int *count);
hipError_t launch_instance(const Instance &inst, const LaunchParams &p, dim3 grid, size_t lds, hipStream_t stream);
hipError_t launch_walk_rays(const LaunchParams &p, const float4 *rays, uint32_t n_rays, uint4 *out, size_t lds, hipStream_t stream);   // diagnostics
hipError_t launch_sqrt_check(unsigned long long *d_mismatches, hipStream_t stream, int which = 0);
hipError_t launch_rcp_probe(unsigned long long *d_hist, hipStream_t stream);
hipError_t launch_eval_parity(int op, const float *in, float *out, size_t n, hipStream_t stream);
hipError_t prepare_parity();    // raise the dynamic-LDS limit (called once per context)
hipError_t prepare_fast();

}  // namespace rt
