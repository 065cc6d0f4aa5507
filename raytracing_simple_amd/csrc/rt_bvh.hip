// rt_bvh.hip -- the hierarchy of large scenes (rt_device.h BvhTables), built on the device behind the scene tables:
// the build kernel and the host function that sizes and launches it.  Walked by rt_walk.inc.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_internal.h"

#ifndef RT_DIAGNOSTICS
#define RT_DIAGNOSTICS 0
#endif

namespace {

// The hierarchy of rt_device.h BvhTables from the raw records, ONE workgroup of 1024 threads:
//   1. a sphere stays outside the tree ("always" list, scene order kept) unless its radius and centre are finite and
//      |rad| <= r_cut (the host derives r_cut from the median radius and the scene's extent: ground planes, walls and lights the
//      size of the scene would blow up every box above them);
//   2. the tree's shape is fixed: leaves of kBvhLeaf spheres, leaf ranges split in the middle, written as sibling pairs;
//   3. who sits in which leaf is decided top-down: the spheres of a node are sorted along the longest axis of the box
//      of their centres, the left child takes the first half of the node's leaves (a median split by count; one
//      bitonic sort per level in LDS).  Against sorting once along a Morton curve this cuts the surface-area cost of
//      the tree by a third (tools/tree_quality.py), which is what the walk pays for;
//   4. boxes are rounded outwards; every node also carries the lowest scene index below it.
// Host and device agree on the counts because they apply the same test to the same bits (bvh_outside).
__host__ __device__ inline bool bvh_outside(float rad, float px, float py, float pz, float r_cut) {
    const float big = 3.0e38f;
    const bool finite = (fabsf(rad) <= big) && (fabsf(px) <= big) && (fabsf(py) <= big) && (fabsf(pz) <= big);   // false for NaN
    return !(finite && fabsf(rad) <= r_cut);
}
// Half the width of a sphere's box.  The walk grows every box by a per-ray pad min(sqrt(eps), eps / 2 r_min) (rt_walk.inc.h), r_min the
// smallest radius in the tree -- so ONE zero-radius record (the .scn loader's doubling puts N of them at the origin, Utility.cpp:120,154)
// would turn that into sqrt(eps) for every box of the tree and every ray: complex.scn's walk took 18 % more pair steps and 46 % more leaf
// visits for its 783 phantoms (profiles/r06_reference_scenes.jsonl).  Instead the header's r_min is the smallest REGULAR radius R
// (|rad| >= r_floor = 1/16 of the median radius) and the boxes of the smaller spheres are grown by g = R / 2 at build time: for a sphere of
// radius r_s < R the point X of the derivation lies within r_s + min(sqrt(eps), eps / 2 r_s) <= r_s + sqrt(eps) of its centre, and
// g + min(s, s^2 / 2R) >= s for every s = sqrt(eps) >= 0 (the difference s - s^2 / 2R peaks at s = R with R / 2; beyond s = 2R the pad is s itself).
// A tree without a regular sphere keeps the true minimum and g = 0, as before.
__host__ __device__ inline float bvh_half_width(float ar, float r_floor, float g) { return ar >= r_floor ? ar : ar + g; }
__device__ inline unsigned bvh_ordered(float f) {          // unsigned order = float order
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float bvh_unordered(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
__device__ inline uint32_t bvh_complaints(uint32_t bad) { return (bad < 0xffu ? bad : 0xffu) << 24; }     // (top byte of the header's last word; bits 16..23: levels of a tree shaped on the device)
__device__ inline float bvh_down(float v) { return v - (fabsf(v) * 0x1p-22f + 1e-30f); }
__device__ inline float bvh_up(float v) { return v + (fabsf(v) * 0x1p-22f + 1e-30f); }

// the node of `level` (or the shallower single-leaf node) that leaf `leaf` lies in, numbered left to right at that level;
// bit 31 set: the node is one leaf (nothing to order)
__device__ inline uint32_t bvh_node_of(uint32_t leaf, uint32_t level, uint32_t n_leaves) {
    uint32_t a = 0, b = n_leaves, rank = 0, s2 = 0;
    for (; s2 < level; ++s2) {
        if (b - a == 1) break;
        const uint32_t mid = (a + b) / 2;
        if (leaf < mid) { b = mid; rank = 2 * rank; }
        else { a = mid; rank = 2 * rank + 1; }
    }
    return (rank << (level - s2)) | ((b - a == 1) ? 0x80000000u : 0u);
}

__global__ void __launch_bounds__(1024) rt_bvh_build_kernel(const rt_sphere *sph, const uint8_t *dup, uint32_t n, float r_cut, float r_floor, uint32_t n_always,
                                                            uint32_t n_tree, uint32_t n_pad, float4 *blob) {
    extern __shared__ unsigned long long s_keys[];          // n_pad sort keys, later 2 float4 per leaf
    __shared__ unsigned s_rmin, s_rmax, s_rmin_all;         // (s_rmin: over the regular radii, >= r_floor; s_rmin_all: over all of the tree's)
    __shared__ uint32_t s_wave_a[16], s_wave_t[16], s_base_a, s_base_t, s_bad;
    const unsigned tid = threadIdx.x, wave = tid >> 6;
    const uint32_t n_leaves = (n_tree + rt::kBvhLeaf - 1) / rt::kBvhLeaf;
    const uint32_t n_slots = n_always + rt::kBvhLeaf * n_leaves;
    float4 *hdr = blob, *slots = blob + rt::bvh_slots_at();
    uint32_t *index = reinterpret_cast<uint32_t *>(blob + rt::bvh_index_at(n_slots));
    if (tid == 0) { s_rmin = 0xffffffffu; s_rmin_all = 0xffffffffu; s_rmax = 0u; s_base_a = 0; s_base_t = 0; s_bad = 0; }
    for (uint32_t i = tid; i < n_pad; i += 1024) s_keys[i] = ~0ull;
    __syncthreads();
    // ---- 1. radius range of the tree's spheres ----
    for (uint32_t i = tid; i < n; i += 1024) {
        const float *r = reinterpret_cast<const float *>(sph + i);
        if (!(dup && dup[i]) && !bvh_outside(r[0], r[1], r[2], r[3], r_cut)) {
            const float ar = fabsf(r[0]);
            atomicMin(&s_rmin_all, __float_as_uint(ar));
            if (ar >= r_floor) atomicMin(&s_rmin, __float_as_uint(ar));
            atomicMax(&s_rmax, __float_as_uint(ar));
        }
    }
    // ---- 2. the always list in scene order, the tree's spheres in scene order for a start (ballot prefix per 1024 records) ----
    for (uint32_t i0 = 0; i0 < n; i0 += 1024) {
        const uint32_t i = i0 + tid;
        bool out = false, in = false;
        float rad = 0.f, px = 0.f, py = 0.f, pz = 0.f;
        if (i < n) {
            const float *r = reinterpret_cast<const float *>(sph + i);
            rad = r[0]; px = r[1]; py = r[2]; pz = r[3];
            const bool repeated = dup && dup[i];                // (a record an earlier one repeats bit for bit: in neither list, mark_duplicates)
            out = !repeated && bvh_outside(rad, px, py, pz, r_cut);
            in = !repeated && !out;
        }
        const unsigned long long ma = __builtin_amdgcn_ballot_w64(out), mt = __builtin_amdgcn_ballot_w64(in);
        const uint32_t before_a = __builtin_amdgcn_mbcnt_hi((uint32_t)(ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ma, 0u));
        const uint32_t before_t = __builtin_amdgcn_mbcnt_hi((uint32_t)(mt >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mt, 0u));
        if ((tid & 63) == 0) { s_wave_a[wave] = (uint32_t)__popcll(ma); s_wave_t[wave] = (uint32_t)__popcll(mt); }
        __syncthreads();
        uint32_t off_a = s_base_a, off_t = s_base_t;
        for (unsigned k = 0; k < wave; ++k) { off_a += s_wave_a[k]; off_t += s_wave_t[k]; }
        if (out) {
            const uint32_t j = off_a + before_a;
            if (j < n_always) {
                slots[j] = make_float4(px, py, pz, rad * rad);
                index[j] = i;
                const float *r = reinterpret_cast<const float *>(sph + i);
                blob[rt::bvh_emis_at(n_leaves, n_slots) + j] = make_float4(r[4], r[5], r[6], r[10]);
                blob[rt::bvh_colr_at(n_leaves, n_slots) + j] = make_float4(r[7], r[8], r[9], rad);
            }
            else atomicAdd(&s_bad, 1u);
        }
        if (in) {
            const uint32_t j = off_t + before_t;
            if (j < n_tree) s_keys[j] = i;
            else atomicAdd(&s_bad, 1u);
        }
        __syncthreads();
        if (tid == 0)
            for (int k = 0; k < 16; ++k) { s_base_a += s_wave_a[k]; s_base_t += s_wave_t[k]; }
        __syncthreads();
    }
    // ---- 2b. order: level by level, the spheres of every node are sorted along the longest axis of the box of
    //      their centres; the node's left child then takes the first half of its leaves, the right child the rest
    //      (the tree's shape -- leaf ranges split in the middle -- is fixed; this decides who sits where).  One
    //      bitonic sort per level over keys  node << 50 | ordered(coordinate) << 18 | scene index: segments never mix.
    uint32_t depth = 0;
    while ((1u << depth) < n_leaves) depth += 1;
    unsigned *s_box = reinterpret_cast<unsigned *>(s_keys + n_pad);      // per node of the level: lo[3], hi[3] (ordered), then the axis
    constexpr unsigned long long kIdxMask = (1ull << 18) - 1;
    for (uint32_t level = 0; level < depth; ++level) {
        const uint32_t n_level = 1u << level;
        for (uint32_t q = tid; q < n_level; q += 1024) {
            s_box[7 * q + 0] = s_box[7 * q + 1] = s_box[7 * q + 2] = 0xffffffffu;
            s_box[7 * q + 3] = s_box[7 * q + 4] = s_box[7 * q + 5] = 0u;
        }
        __syncthreads();
        for (uint32_t j = tid; j < n_tree; j += 1024) {
            const uint32_t node = bvh_node_of(j / rt::kBvhLeaf, level, n_leaves);
            if (node & 0x80000000u) continue;
            const uint32_t rank = node;
            const float *r = reinterpret_cast<const float *>(sph + (uint32_t)(s_keys[j] & kIdxMask));
            for (int a3 = 0; a3 < 3; ++a3) {
                atomicMin(&s_box[7 * rank + a3], bvh_ordered(r[1 + a3]));
                atomicMax(&s_box[7 * rank + 3 + a3], bvh_ordered(r[1 + a3]));
            }
        }
        __syncthreads();
        for (uint32_t q = tid; q < n_level; q += 1024) {
            float ext[3];
            for (int a3 = 0; a3 < 3; ++a3) {
                const unsigned ulo = s_box[7 * q + a3], uhi = s_box[7 * q + 3 + a3];
                ext[a3] = uhi >= ulo ? bvh_unordered(uhi) - bvh_unordered(ulo) : 0.f;     // (a node nobody touched: no extent)
            }
            unsigned axis = 0;
            if (ext[1] > ext[axis]) axis = 1;
            if (ext[2] > ext[axis]) axis = 2;
            s_box[7 * q + 6] = axis;
        }
        __syncthreads();
        for (uint32_t j = tid; j < n_tree; j += 1024) {
            const uint32_t node = bvh_node_of(j / rt::kBvhLeaf, level, n_leaves), rank = node & 0x7fffffffu;
            const uint32_t ix = (uint32_t)(s_keys[j] & kIdxMask);
            const float *r = reinterpret_cast<const float *>(sph + ix);
            const unsigned axis = s_box[7 * rank + 6];
            const float c3 = axis == 0 ? r[1] : (axis == 1 ? r[2] : r[3]);
            const unsigned coord = (node & 0x80000000u) ? j : bvh_ordered(c3);   // a node of one leaf keeps its order
            s_keys[j] = ((unsigned long long)rank << 50) | ((unsigned long long)coord << 18) | ix;
        }
        __syncthreads();
        for (uint32_t k = 2; k <= n_pad; k <<= 1) {             // bitonic sort, ascending (the padding keys stay behind)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < n_pad; i += 1024) {
                    const uint32_t l = i ^ j;
                    if (l > i) {
                        const unsigned long long ka = s_keys[i], kb = s_keys[l];
                        if ((ka > kb) == ((i & k) == 0)) { s_keys[i] = kb; s_keys[l] = ka; }
                    }
                }
                __syncthreads();
            }
        }
    }
    // ---- 3. records in leaf order; padding records never hit (NaN centre: every comparison of the test is false) ----
    const float qnan = __uint_as_float(0x7fc00000u);
    for (uint32_t j = tid; j < rt::kBvhLeaf * n_leaves; j += 1024) {
        const uint32_t ix = j < n_tree ? (uint32_t)(s_keys[j] & kIdxMask) : 0xffffffffu;
        if (ix != 0xffffffffu) {
            const float *r = reinterpret_cast<const float *>(sph + ix);
            slots[n_always + j] = make_float4(r[1], r[2], r[3], r[0] * r[0]);
            blob[rt::bvh_emis_at(n_leaves, n_slots) + n_always + j] = make_float4(r[4], r[5], r[6], r[10]);
            blob[rt::bvh_colr_at(n_leaves, n_slots) + n_always + j] = make_float4(r[7], r[8], r[9], r[0]);
        } else {
            slots[n_always + j] = make_float4(qnan, qnan, qnan, qnan);
            blob[rt::bvh_emis_at(n_leaves, n_slots) + n_always + j] = make_float4(0.f, 0.f, 0.f, 0.f);
            blob[rt::bvh_colr_at(n_leaves, n_slots) + n_always + j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        index[n_always + j] = ix;
    }
    __threadfence_block();                                  // the index written above is read back below by other threads
    __syncthreads();                                        // the keys are dead: the same LDS now holds the leaf boxes
    float4 *s_leaf = reinterpret_cast<float4 *>(s_keys);
    const bool have_regular = s_rmin != 0xffffffffu;
    const float r_min = __uint_as_float(have_regular ? s_rmin : s_rmin_all), grow = have_regular ? 0.5f * r_min : 0.f;       // (bvh_half_width)
    for (uint32_t leaf = tid; leaf < n_leaves; leaf += 1024) {
        float lo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, hi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
        uint32_t low = 0xffffffffu;
        for (int k = 0; k < rt::kBvhLeaf; ++k) {
            const uint32_t ix = index[n_always + rt::kBvhLeaf * leaf + k];
            if (ix == 0xffffffffu) continue;
            const float *r = reinterpret_cast<const float *>(sph + ix);
            const float ar = bvh_half_width(fabsf(r[0]), r_floor, grow);
            for (int a = 0; a < 3; ++a) {
                lo[a] = fminf(lo[a], bvh_down(r[1 + a] - ar));
                hi[a] = fmaxf(hi[a], bvh_up(r[1 + a] + ar));
            }
            low = ix < low ? ix : low;
        }
        s_leaf[2 * leaf] = make_float4(lo[0], lo[1], lo[2], 0.f);
        s_leaf[2 * leaf + 1] = make_float4(hi[0], hi[1], hi[2], __uint_as_float(low));
    }
    __syncthreads();
    // ---- 4. the header: root box (union of the leaf boxes) and radius range ----
    if (tid == 0) {
        if (n_leaves) {
            float lo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, hi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
            for (uint32_t l = 0; l < n_leaves; ++l) {
                const float4 A = s_leaf[2 * l], B = s_leaf[2 * l + 1];
                lo[0] = fminf(lo[0], A.x); lo[1] = fminf(lo[1], A.y); lo[2] = fminf(lo[2], A.z);
                hi[0] = fmaxf(hi[0], B.x); hi[1] = fmaxf(hi[1], B.y); hi[2] = fmaxf(hi[2], B.z);
            }
            const float cx = 0.5f * lo[0] + 0.5f * hi[0], cy = 0.5f * lo[1] + 0.5f * hi[1], cz = 0.5f * lo[2] + 0.5f * hi[2];
            const float ex = hi[0] - cx, ey = hi[1] - cy, ez = hi[2] - cz;
            hdr[0] = make_float4(cx, cy, cz, sqrtf(ex * ex + ey * ey + ez * ez) * 1.001f);
            const float rmin = r_min, rmax = __uint_as_float(s_rmax);
            hdr[1] = make_float4(rmin, rmax, 1.f / (2.f * rmin), __uint_as_float((n_leaves > 1 ? n_leaves / 2u - 1u : rt::kBvhLeafRef) | bvh_complaints(s_bad)));
        } else {
            hdr[0] = make_float4(0.f, 0.f, 0.f, 0.f);
            hdr[1] = make_float4(0.f, 0.f, 0.f, __uint_as_float(rt::kBvhLeafRef | bvh_complaints(s_bad)));
        }
    }
    // ---- 5. the sibling pairs (rt_device.h BvhTables `pairs`): one thread per inner node ----
    float4 *pairs = blob + rt::bvh_pairs_at(n_slots);
    for (uint32_t m = 1 + tid; m < n_leaves; m += 1024) {
        uint32_t a = 0, b = n_leaves, mid;
        for (;;) {
            mid = (a + b) / 2;
            if (mid == m) break;
            if (m < mid) b = mid;
            else a = mid;
        }
        for (int side = 0; side < 2; ++side) {
            const uint32_t ca = side ? mid : a, cb = side ? b : mid;
            float lo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, hi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
            uint32_t low = 0xffffffffu;
            for (uint32_t l = ca; l < cb; ++l) {
                const float4 A = s_leaf[2 * l], B = s_leaf[2 * l + 1];
                lo[0] = fminf(lo[0], A.x); lo[1] = fminf(lo[1], A.y); lo[2] = fminf(lo[2], A.z);
                hi[0] = fmaxf(hi[0], B.x); hi[1] = fmaxf(hi[1], B.y); hi[2] = fmaxf(hi[2], B.z);
                const uint32_t q = __float_as_uint(B.w);
                low = q < low ? q : low;
            }
            const uint32_t ref = (cb - ca == 1) ? (rt::kBvhLeafRef | ca) : (ca + cb) / 2 - 1;
            pairs[4 * (size_t)(m - 1) + 2 * side] = make_float4(lo[0], lo[1], lo[2], __uint_as_float(ref));
            pairs[4 * (size_t)(m - 1) + 2 * side + 1] = make_float4(hi[0], hi[1], hi[2], __uint_as_float(low));
        }
    }
}

// The same tables with the tree's SHAPE chosen by surface area ON THE DEVICE (device-resident updates, and uploads whose host
// build would stall the caller for milliseconds): ONE workgroup, the same sort per level, but a node's range of leaves is cut
// where  area(left) * leaves(left) + area(right) * leaves(right)  is smallest instead of in the middle.  Cuts fall between WHOLE
// leaves (every leaf full but the last), so the leaf count, the slot count and every offset into the blob are what the host
// computes from the sphere count alone -- nothing is read back.  What the host cannot know travels with the tree: the root's
// pair in the header; the depth is bounded instead (a node at level k may hold at most 2^(depth_cap - k) leaves, so the stacks are
// sized for depth_cap).  Against the host's build by surface area (any cut, three axes): one axis per node (the longest of its
// box), no partial leaves -- the model of round 3 put cuts at whole leaves within 2 % of any cut on C3 (profiles/r03y_*).
//   per level:  keys (node's first leaf | coordinate along the node's axis | scene index) -> bitonic sort -> leaf boxes ->
//               segmented prefix / suffix unions of the leaf boxes inside every node (Hillis-Steele, 7 words per box) ->
//               one thread per cut evaluates its cost, ds_min_u64 per node picks -> the pair record of the cut, the children's
//               ranges, axes and references.  Inner node (a, b) cut at c has its pair at c - 1, as in every other build.
constexpr uint32_t kSahDeviceMaxTree = 1024 * rt::kBvhLeaf;        // 1024 leaves (8192 spheres): a thread per leaf (up to 512 leaves the unions towards both ends of a node are formed at once)
__device__ inline float bvh_area7(const float *b) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}
__device__ inline unsigned bvh_longest(const float *b) {
    const float ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2];
    unsigned axis = 0;
    float e = ex;
    if (ey > e) { axis = 1; e = ey; }
    if (ez > e) axis = 2;
    return axis;
}
__global__ void __launch_bounds__(1024) rt_bvh_build_sah_kernel(const rt_sphere *sph, const uint8_t *dup, uint32_t n, float r_cut, float r_floor, uint32_t n_always, uint32_t n_tree,
                                                                uint32_t n_pad, uint32_t depth_cap, float4 *blob) {
    extern __shared__ unsigned long long s_keys[];          // n_pad sort keys, then the per-leaf arrays below
    __shared__ unsigned s_rmin, s_rmin_all, s_rmax, s_rb[6];
    __shared__ uint32_t s_wave_a[16], s_wave_t[16], s_base_a, s_base_t, s_bad, s_any, s_root, s_levels;
    const unsigned tid = threadIdx.x, wave = tid >> 6;
    const uint32_t L = (n_tree + rt::kBvhLeaf - 1) / rt::kBvhLeaf;
    const uint32_t n_slots = n_always + rt::kBvhLeaf * L;
    float4 *hdr = blob, *slots = blob + rt::bvh_slots_at();
    uint32_t *index = reinterpret_cast<uint32_t *>(blob + rt::bvh_index_at(n_slots));
    float4 *pairs = blob + rt::bvh_pairs_at(n_slots);
    float *s_pre = reinterpret_cast<float *>(s_keys + n_pad);       // [7][L]: lo.xyz, hi.xyz, bits(lowest scene index): unions from the node's first leaf up to this one
    float *s_suf = s_pre + 7 * L;                                   // ... from this leaf to the node's last
    unsigned long long *s_best = reinterpret_cast<unsigned long long *>(s_suf + 7 * L);      // per node (by its first leaf): bits(cost) << 32 | cut
    uint16_t *s_na = reinterpret_cast<uint16_t *>(s_best + L), *s_nb = s_na + L;             // per leaf: the range of leaves of the node it is in
    uint16_t *s_refl = s_nb + L, *s_refr = s_refl + L;                                       // per cut: the references of the pair's two children
    uint8_t *s_side = reinterpret_cast<uint8_t *>(s_refr + L), *s_axis = s_side + L;         // per leaf: its node is the left (0) / right (1) child, or the root (2); per node: the axis
    if (tid == 0) { s_rmin = 0xffffffffu; s_rmin_all = 0xffffffffu; s_rmax = 0u; s_base_a = 0; s_base_t = 0; s_bad = 0; s_root = rt::kBvhLeafRef; s_levels = 0; }
    if (tid < 3) { s_rb[tid] = 0xffffffffu; s_rb[3 + tid] = 0u; }
    for (uint32_t i = tid; i < n_pad; i += 1024) s_keys[i] = ~0ull;
    for (uint32_t l = tid; l < L; l += 1024) { s_na[l] = 0; s_nb[l] = (uint16_t)L; s_side[l] = 2; s_axis[l] = 0; s_refl[l] = s_refr[l] = 0xffffu; }
    __syncthreads();
    // ---- 1. radius range of the tree's spheres, and their box (the root's, whatever the order: it also gives the root's axis) ----
    for (uint32_t i = tid; i < n; i += 1024) {
        const float *r = reinterpret_cast<const float *>(sph + i);
        if (!(dup && dup[i]) && !bvh_outside(r[0], r[1], r[2], r[3], r_cut)) {
            const float ar = fabsf(r[0]);
            atomicMin(&s_rmin_all, __float_as_uint(ar));
            if (ar >= r_floor) atomicMin(&s_rmin, __float_as_uint(ar));
            atomicMax(&s_rmax, __float_as_uint(ar));
            const float hw = bvh_half_width(ar, r_floor, 8.f * r_floor);        // (the growth of a small sphere's box is R / 2 <= median / 2 = 8 r_floor: R is not known yet)
            for (int a3 = 0; a3 < 3; ++a3) {
                atomicMin(&s_rb[a3], bvh_ordered(bvh_down(r[1 + a3] - hw)));
                atomicMax(&s_rb[3 + a3], bvh_ordered(bvh_up(r[1 + a3] + hw)));
            }
        }
    }
    // ---- 2. the always list in scene order, the tree's spheres in scene order for a start (as in rt_bvh_build_kernel) ----
    for (uint32_t i0 = 0; i0 < n; i0 += 1024) {
        const uint32_t i = i0 + tid;
        bool out = false, in = false;
        float rad = 0.f, px = 0.f, py = 0.f, pz = 0.f;
        if (i < n) {
            const float *r = reinterpret_cast<const float *>(sph + i);
            rad = r[0]; px = r[1]; py = r[2]; pz = r[3];
            const bool repeated = dup && dup[i];                // (a record an earlier one repeats bit for bit: in neither list, mark_duplicates)
            out = !repeated && bvh_outside(rad, px, py, pz, r_cut);
            in = !repeated && !out;
        }
        const unsigned long long ma = __builtin_amdgcn_ballot_w64(out), mt = __builtin_amdgcn_ballot_w64(in);
        const uint32_t before_a = __builtin_amdgcn_mbcnt_hi((uint32_t)(ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ma, 0u));
        const uint32_t before_t = __builtin_amdgcn_mbcnt_hi((uint32_t)(mt >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mt, 0u));
        if ((tid & 63) == 0) { s_wave_a[wave] = (uint32_t)__popcll(ma); s_wave_t[wave] = (uint32_t)__popcll(mt); }
        __syncthreads();
        uint32_t off_a = s_base_a, off_t = s_base_t;
        for (unsigned k = 0; k < wave; ++k) { off_a += s_wave_a[k]; off_t += s_wave_t[k]; }
        if (out) {
            const uint32_t j = off_a + before_a;
            if (j < n_always) {
                slots[j] = make_float4(px, py, pz, rad * rad);
                index[j] = i;
                const float *r = reinterpret_cast<const float *>(sph + i);
                blob[rt::bvh_emis_at(L, n_slots) + j] = make_float4(r[4], r[5], r[6], r[10]);
                blob[rt::bvh_colr_at(L, n_slots) + j] = make_float4(r[7], r[8], r[9], rad);
            }
            else atomicAdd(&s_bad, 1u);
        }
        if (in) {
            const uint32_t j = off_t + before_t;
            if (j < n_tree) s_keys[j] = i;
            else atomicAdd(&s_bad, 1u);
        }
        __syncthreads();
        if (tid == 0)
            for (int k = 0; k < 16; ++k) { s_base_a += s_wave_a[k]; s_base_t += s_wave_t[k]; }
        __syncthreads();
    }
    if (tid == 0) {
        float rb[6];
        for (int k = 0; k < 6; ++k) rb[k] = bvh_unordered(s_rb[k]);
        s_axis[0] = (uint8_t)bvh_longest(rb);
    }
    __syncthreads();
    // ---- 2b. level by level: order every node's spheres along its axis, cut its range of leaves by surface area ----
    constexpr unsigned long long kIdxMask = (1ull << 18) - 1;
    const bool both_at_once = L <= 512u;                 // threads 0..511 form the unions towards a node's first leaf while 512..1023 form those towards its last
    for (uint32_t level = 0;; ++level) {
        for (uint32_t j = tid; j < n_tree; j += 1024) {
            const uint32_t l = j / rt::kBvhLeaf, a = s_na[l];
            const bool single = (uint32_t)s_nb[l] - a == 1u;
            const uint32_t ix = (uint32_t)(s_keys[j] & kIdxMask);
            const float *r = reinterpret_cast<const float *>(sph + ix);
            const unsigned axis = s_axis[a];
            const float c3 = axis == 0 ? r[1] : (axis == 1 ? r[2] : r[3]);
            const unsigned coord = single ? j : bvh_ordered(c3);        // a node of one leaf keeps its order
            s_keys[j] = ((unsigned long long)a << 50) | ((unsigned long long)coord << 18) | ix;
        }
        for (uint32_t l = tid; l < L; l += 1024) s_best[l] = ~0ull;
        if (tid == 0) s_any = 0;
        __syncthreads();
        for (uint32_t k = 2; k <= n_pad; k <<= 1) {             // bitonic sort, ascending (the padding keys stay behind)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < n_pad; i += 1024) {
                    const uint32_t l = i ^ j;
                    if (l > i) {
                        const unsigned long long ka = s_keys[i], kb = s_keys[l];
                        if ((ka > kb) == ((i & k) == 0)) { s_keys[i] = kb; s_keys[l] = ka; }
                    }
                }
                __syncthreads();
            }
        }
        // leaf boxes (rounded outwards) and the lowest scene index in the leaf: the start of both scans
        if (tid < L) {
            float lo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, hi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
            uint32_t low = 0xffffffffu;
            for (int k = 0; k < rt::kBvhLeaf; ++k) {
                const uint32_t j = rt::kBvhLeaf * tid + (uint32_t)k;
                if (j >= n_tree) break;
                const uint32_t ix = (uint32_t)(s_keys[j] & kIdxMask);
                const float *r = reinterpret_cast<const float *>(sph + ix);
                const float ar = bvh_half_width(fabsf(r[0]), r_floor, s_rmin != 0xffffffffu ? 0.5f * __uint_as_float(s_rmin) : 0.f);
                for (int a3 = 0; a3 < 3; ++a3) {
                    lo[a3] = fminf(lo[a3], bvh_down(r[1 + a3] - ar));
                    hi[a3] = fmaxf(hi[a3], bvh_up(r[1 + a3] + ar));
                }
                low = ix < low ? ix : low;
            }
            for (int a3 = 0; a3 < 3; ++a3) {
                s_pre[a3 * L + tid] = s_suf[a3 * L + tid] = lo[a3];
                s_pre[(3 + a3) * L + tid] = s_suf[(3 + a3) * L + tid] = hi[a3];
            }
            s_pre[6 * L + tid] = s_suf[6 * L + tid] = __uint_as_float(low);
        }
        __syncthreads();
        // segmented scans: s_pre[.][l] = union of the node's leaves up to l, s_suf[.][l] = from l on
        for (int pass = 0; pass < (both_at_once ? 1 : 2); ++pass) {
            const uint32_t scan_l = both_at_once ? (tid & 511u) : tid;
            const bool scan_up = both_at_once ? tid >= 512u : pass == 1;
            float *s_mine = scan_up ? s_suf : s_pre;
            const bool mine = scan_l < L;
            const uint32_t a = mine ? s_na[scan_l] : 0u, b = mine ? s_nb[scan_l] : 0u;
            float v[7];
            if (mine)
                for (int k = 0; k < 7; ++k) v[k] = s_mine[k * L + scan_l];
            for (uint32_t dist = 1; dist < L; dist <<= 1) {
                const uint32_t other = scan_up ? scan_l + dist : scan_l - dist;
                const bool ok = mine && (scan_up ? other < b : (scan_l >= dist && other >= a));
                if (ok) {
                    for (int k = 0; k < 3; ++k) v[k] = fminf(v[k], s_mine[k * L + other]);
                    for (int k = 3; k < 6; ++k) v[k] = fmaxf(v[k], s_mine[k * L + other]);
                    const uint32_t q = __float_as_uint(s_mine[6 * L + other]), w = __float_as_uint(v[6]);
                    v[6] = __uint_as_float(q < w ? q : w);
                }
                __syncthreads();
                if (ok)
                    for (int k = 0; k < 7; ++k) s_mine[k * L + scan_l] = v[k];
                __syncthreads();
            }
        }
        // every cut's cost; the cheapest per node (ties: the lower cut)
        if (tid < L) {
            const uint32_t c = tid, a = s_na[c], b = s_nb[c];
            const uint32_t room = depth_cap > level + 1u ? depth_cap - level - 1u : 0u;
            const uint32_t most = room >= 16u ? 0xffffu : (1u << room);         // leaves a child may hold and still end within depth_cap
            if (c > a && c - a <= most && b - c <= most) {
                float lb[6], rb[6];
                for (int k = 0; k < 6; ++k) { lb[k] = s_pre[k * L + c - 1]; rb[k] = s_suf[k * L + c]; }
                const float cost = bvh_area7(lb) * (float)(c - a) + bvh_area7(rb) * (float)(b - c);
                atomicMin(&s_best[a], ((unsigned long long)__float_as_uint(cost) << 32) | c);
            }
        }
        __syncthreads();
        if (tid < L) {
            const uint32_t l = tid, a = s_na[l], b = s_nb[l];
            if (b - a > 1u) {
                const unsigned long long k = s_best[a];
                const uint32_t c = k == ~0ull ? a + (b - a) / 2u : (uint32_t)k;
                if (l == c) {
                    float lb[7], rb[7];
                    for (int q = 0; q < 7; ++q) { lb[q] = s_pre[q * L + c - 1]; rb[q] = s_suf[q * L + c]; }
                    pairs[4 * (size_t)(c - 1) + 0] = make_float4(lb[0], lb[1], lb[2], 0.f);
                    pairs[4 * (size_t)(c - 1) + 1] = make_float4(lb[3], lb[4], lb[5], lb[6]);
                    pairs[4 * (size_t)(c - 1) + 2] = make_float4(rb[0], rb[1], rb[2], 0.f);
                    pairs[4 * (size_t)(c - 1) + 3] = make_float4(rb[3], rb[4], rb[5], rb[6]);
                    s_refl[c] = (uint16_t)(c - a == 1u ? (rt::kBvhLeafRef | a) : 0xffffu);
                    s_refr[c] = (uint16_t)(b - c == 1u ? (rt::kBvhLeafRef | c) : 0xffffu);
                    const unsigned sd = s_side[l];
                    if (sd == 2u) s_root = c - 1u;
                    else if (sd == 0u) s_refl[b] = (uint16_t)(c - 1u);           // (a left child's parent was cut at the child's end, a right child's at its start)
                    else s_refr[a] = (uint16_t)(c - 1u);
                    s_axis[a] = (uint8_t)bvh_longest(lb);
                    s_axis[c] = (uint8_t)bvh_longest(rb);
                    s_levels = level + 1u;
                }
                if (l < c) { s_nb[l] = (uint16_t)c; s_side[l] = 0; if (c - a > 1u) s_any = 1u; }
                else { s_na[l] = (uint16_t)c; s_side[l] = 1; if (b - c > 1u) s_any = 1u; }
            }
        }
        __syncthreads();
        if (s_any == 0u) break;
        __syncthreads();            // (s_any is cleared at the top of the next level)
    }
    // ---- 3. records in leaf order; padding records never hit (NaN centre: every comparison of the test is false) ----
    const float qnan = __uint_as_float(0x7fc00000u);
    for (uint32_t j = tid; j < rt::kBvhLeaf * L; j += 1024) {
        const uint32_t ix = j < n_tree ? (uint32_t)(s_keys[j] & kIdxMask) : 0xffffffffu;
        if (ix != 0xffffffffu) {
            const float *r = reinterpret_cast<const float *>(sph + ix);
            slots[n_always + j] = make_float4(r[1], r[2], r[3], r[0] * r[0]);
            blob[rt::bvh_emis_at(L, n_slots) + n_always + j] = make_float4(r[4], r[5], r[6], r[10]);
            blob[rt::bvh_colr_at(L, n_slots) + n_always + j] = make_float4(r[7], r[8], r[9], r[0]);
        } else {
            slots[n_always + j] = make_float4(qnan, qnan, qnan, qnan);
            blob[rt::bvh_emis_at(L, n_slots) + n_always + j] = make_float4(0.f, 0.f, 0.f, 0.f);
            blob[rt::bvh_colr_at(L, n_slots) + n_always + j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        index[n_always + j] = ix;
    }
    // ---- 4. the pairs' references (each child entered its own when it was cut), the header ----
    __threadfence_block();
    __syncthreads();
    for (uint32_t c = 1 + tid; c < L; c += 1024) {
        float4 r0 = pairs[4 * (size_t)(c - 1) + 0], r2 = pairs[4 * (size_t)(c - 1) + 2];
        r0.w = __uint_as_float((uint32_t)s_refl[c]);
        r2.w = __uint_as_float((uint32_t)s_refr[c]);
        pairs[4 * (size_t)(c - 1) + 0] = r0;
        pairs[4 * (size_t)(c - 1) + 2] = r2;
    }
    if (tid == 0) {
        const uint32_t last = (L > 1 ? s_root : rt::kBvhLeafRef) | ((s_levels & 0xffu) << 16) | bvh_complaints(s_bad);
        if (L) {
            float rb[6];
            for (int k = 0; k < 6; ++k) rb[k] = bvh_unordered(s_rb[k]);
            const float cx = 0.5f * rb[0] + 0.5f * rb[3], cy = 0.5f * rb[1] + 0.5f * rb[4], cz = 0.5f * rb[2] + 0.5f * rb[5];
            const float ex = rb[3] - cx, ey = rb[4] - cy, ez = rb[5] - cz;
            hdr[0] = make_float4(cx, cy, cz, sqrtf(ex * ex + ey * ey + ez * ez) * 1.001f);
            const float rmin = __uint_as_float(s_rmin != 0xffffffffu ? s_rmin : s_rmin_all), rmax = __uint_as_float(s_rmax);
            hdr[1] = make_float4(rmin, rmax, 1.f / (2.f * rmin), __uint_as_float(last));
        } else {
            hdr[0] = make_float4(0.f, 0.f, 0.f, 0.f);
            hdr[1] = make_float4(0.f, 0.f, 0.f, __uint_as_float(last));
        }
    }
}

#if RT_DIAGNOSTICS
// ---- Two arms of the A/B on the walk that reads its tables from HBM / L2 (measured, not adopted) --
// they exist in the diagnostics library only: the product's trees keep the builders' numbering and carry no packed table. ----
// The TOP of the tree to the front of the pair table (BvhTables::n_top), for the walk that reads its tables from HBM / L2 and stages only
// this much of them in LDS (rt_walk.inc.h RT_OPT_TOP_PAIRS).  Every builder numbers the pairs by their leaves (pair of a node = first leaf
// of its right child - 1), which is what lets the host lay the blob out before the tree exists; that order scatters the top levels over
// the whole table.  This pass renumbers AFTER any builder, host or device, on the stream behind it: the first `n_top` pairs in
// breadth-first order from the root take the places [0, n_top) (the root's pair becomes pair 0), every other pair keeps its place in the
// builders' order behind them, the child references and the header's root are rewritten -- a permutation of the table, so every offset
// into the blob, the leaf numbers and the stack depth are what they were.  n_top = min(top_max, pairs): a breadth-first walk from the root
// reaches every pair, so the host knows the count without reading anything back.  ONE workgroup; `scratch`: 4 float4 per pair, then one
// u16 per pair (the old -> new map).
__global__ void __launch_bounds__(1024) rt_bvh_promote_kernel(float4 *blob, uint32_t n_slots, uint32_t n_leaves, uint32_t top_max, float4 *scratch) {
    __shared__ uint16_t s_bfs[256], s_sorted[256], s_rank_bfs[256], s_refs[2 * 256];
    __shared__ uint32_t s_count;
    const unsigned tid = threadIdx.x;
    const uint32_t n_pairs = n_leaves ? n_leaves - 1u : 0u;
    float4 *hdr = blob, *pairs = blob + rt::bvh_pairs_at(n_slots);
    uint16_t *map = reinterpret_cast<uint16_t *>(scratch + 4 * (size_t)n_pairs);
    const uint32_t last = __float_as_uint(hdr[1].w), root = last & 0xffffu;
    if (n_pairs == 0u || (root & rt::kBvhLeafRef) || top_max == 0u) return;
    if (top_max > 256u) top_max = 256u;
    // ---- breadth-first from the root, level by level: the level's references read by its threads, appended in order by one ----
    if (tid == 0) { s_bfs[0] = (uint16_t)root; s_count = 1u; }
    __syncthreads();
    uint32_t lo = 0u;
    for (;;) {
        const uint32_t hi = s_count;
        if (lo >= hi || hi >= top_max) break;
        if (lo + tid < hi) {
            const uint32_t p = s_bfs[lo + tid];
            s_refs[2 * tid] = (uint16_t)(__float_as_uint(pairs[4 * (size_t)p].w) & 0xffffu);
            s_refs[2 * tid + 1] = (uint16_t)(__float_as_uint(pairs[4 * (size_t)p + 2].w) & 0xffffu);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t cnt = hi;
            for (uint32_t k = 0; k < 2u * (hi - lo) && cnt < top_max; ++k) {
                const uint32_t r = s_refs[k];
                if (!(r & rt::kBvhLeafRef)) s_bfs[cnt++] = (uint16_t)r;
            }
            s_count = cnt;
        }
        lo = hi;
        __syncthreads();
    }
    const uint32_t T = s_count;
    // ---- the promoted pairs sorted by their old number (rank by counting: they are distinct), and each one's breadth-first place ----
    if (tid < T) {
        const uint32_t mine = s_bfs[tid];
        uint32_t rank = 0u;
        for (uint32_t j = 0; j < T; ++j) rank += s_bfs[j] < mine ? 1u : 0u;
        s_sorted[rank] = (uint16_t)mine;
        s_rank_bfs[rank] = (uint16_t)tid;
    }
    __syncthreads();
    // ---- old -> new for every pair: a promoted pair goes to its breadth-first place, any other moves up by the promoted pairs behind it ----
    for (uint32_t p = tid; p < n_pairs; p += 1024) {
        uint32_t a = 0u, b = T;                 // first index whose entry is >= p
        while (a < b) {
            const uint32_t m = (a + b) / 2u;
            if (s_sorted[m] < p) a = m + 1u;
            else b = m;
        }
        const bool top = a < T && s_sorted[a] == p;
        map[p] = (uint16_t)(top ? s_rank_bfs[a] : T + p - a);
    }
    __threadfence_block();
    __syncthreads();
    // ---- the permuted table into the scratch copy (references rewritten), and back ----
    for (uint32_t p = tid; p < n_pairs; p += 1024) {
        float4 r0 = pairs[4 * (size_t)p], r1 = pairs[4 * (size_t)p + 1], r2 = pairs[4 * (size_t)p + 2], r3 = pairs[4 * (size_t)p + 3];
        const uint32_t c0 = __float_as_uint(r0.w) & 0xffffu, c1 = __float_as_uint(r2.w) & 0xffffu;
        r0.w = __uint_as_float((c0 & rt::kBvhLeafRef) ? c0 : (uint32_t)map[c0]);
        r2.w = __uint_as_float((c1 & rt::kBvhLeafRef) ? c1 : (uint32_t)map[c1]);
        float4 *q = scratch + 4 * (size_t)map[p];
        q[0] = r0; q[1] = r1; q[2] = r2; q[3] = r3;
    }
    __threadfence_block();
    __syncthreads();
    for (size_t i = tid; i < 4 * (size_t)n_pairs; i += 1024) pairs[i] = scratch[i];
    if (tid == 0) hdr[1].w = __uint_as_float((last & 0xffff0000u) | (uint32_t)map[root]);
}

// The PACKED pair table (rt_device.h BvhTables::packed_at) from the finished pairs, behind any builder and the promotion above, on the same stream:
// per pair 32 bytes -- the two child boxes as twelve 16-bit grid coordinates in the frame of the ROOT box (plane = r0 + q * scale per axis),
// the two references, the two lowest scene indices >> kBvhLowShift.  Every box of the tree lies inside the root's (unions of unions, exact), so
// 16 bits per axis always cover it; a low plane is rounded DOWN to the grid and then one whole cell further, a high plane up and one cell further
// (the quotient's own rounding is a hundredth of a cell), clamped to the frame's ends, which hold the root box by construction: the packed box
// contains the stored one, and culling with it stays one-sided.  What it costs is tightness: a cell is 1 / 65 535 of the root box's extent.
// One thread per pair; every thread forms the frame from the root pair itself (two cached reads), thread 0 of the grid stores it.
__global__ void __launch_bounds__(256) rt_bvh_pack_pairs_kernel(const float4 *blob, uint32_t n_slots, uint32_t n_leaves, float4 *packed) {
    const uint32_t n_pairs = n_leaves ? n_leaves - 1u : 0u;
    const float4 *pairs = blob + rt::bvh_pairs_at(n_slots);
    const uint32_t root = __float_as_uint(blob[1].w) & 0xffffu;
    if (n_pairs == 0u || (root & rt::kBvhLeafRef)) return;
    const float4 ra = pairs[4 * (size_t)root], rb = pairs[4 * (size_t)root + 1], rc = pairs[4 * (size_t)root + 2], rd = pairs[4 * (size_t)root + 3];
    const float r0[3] = { fminf(ra.x, rc.x), fminf(ra.y, rc.y), fminf(ra.z, rc.z) };
    const float r1[3] = { fmaxf(rb.x, rd.x), fmaxf(rb.y, rd.y), fmaxf(rb.z, rd.z) };
    float scale[3], inv_scale[3];
    for (int a = 0; a < 3; ++a) {
        const float ext = r1[a] - r0[a];
        scale[a] = ext > 0.f ? (ext / 65535.f) * 1.000001f : 0.f;          // (r0 + 65535 * scale >= the root's high plane)
        inv_scale[a] = scale[a] > 0.f ? 1.f / scale[a] : 0.f;
    }
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (gid == 0u) {
        packed[0] = make_float4(r0[0], r0[1], r0[2], 0.f);
        packed[1] = make_float4(scale[0], scale[1], scale[2], 0.f);
    }
    uint4 *out = reinterpret_cast<uint4 *>(packed + 2);
    for (uint32_t p = gid; p < n_pairs; p += gridDim.x * 256u) {
        const float4 A0 = pairs[4 * (size_t)p], B0 = pairs[4 * (size_t)p + 1], A1 = pairs[4 * (size_t)p + 2], B1 = pairs[4 * (size_t)p + 3];
        auto down = [&](float v, int a) -> uint32_t {
            const float t = floorf((v - r0[a]) * inv_scale[a]);
            const float q = t - 1.f;                                        // (NaN / negative / beyond the frame: the ends)
            return q > 0.f ? (q < 65535.f ? (uint32_t)q : 65535u) : 0u;
        };
        auto up = [&](float v, int a) -> uint32_t {
            const float t = ceilf((v - r0[a]) * inv_scale[a]);
            const float q = t + 1.f;
            return q < 65535.f ? (q > 0.f ? (uint32_t)q : 0u) : 65535u;
        };
        auto low16 = [](float w) -> uint32_t {
            const uint32_t q = __float_as_uint(w) >> rt::kBvhLowShift;
            return q < 0xffffu ? q : 0xffffu;
        };
        const uint32_t l0x = down(A0.x, 0), l0y = down(A0.y, 1), l0z = down(A0.z, 2), h0x = up(B0.x, 0), h0y = up(B0.y, 1), h0z = up(B0.z, 2);
        const uint32_t l1x = down(A1.x, 0), l1y = down(A1.y, 1), l1z = down(A1.z, 2), h1x = up(B1.x, 0), h1y = up(B1.y, 1), h1z = up(B1.z, 2);
        const uint32_t c0 = __float_as_uint(A0.w) & 0xffffu, c1 = __float_as_uint(A1.w) & 0xffffu;
        out[2 * (size_t)p] = make_uint4(l0x | (l0y << 16), l0z | (h0x << 16), h0y | (h0z << 16), l1x | (l1y << 16));
        out[2 * (size_t)p + 1] = make_uint4(l1z | (h1x << 16), h1y | (h1z << 16), c0 | (c1 << 16), low16(B0.w) | (low16(B1.w) << 16));
    }
}

#endif   // RT_DIAGNOSTICS

// Trees beyond what one workgroup sorts in LDS (more than 8192 spheres in the tree): the same tables from the host
// mirror of the records -- same split, same top-down median ordering, same leaves, same sibling pairs, boxes rounded
// outwards the same way -- written into a page-locked buffer and
// copied on `stream`.  Milliseconds of host time per build for scenes of this size; nothing is waited for.
constexpr uint32_t kDeviceBuildMax = 8192;

inline float host_down(float v) { return v - (fabsf(v) * 0x1p-22f + 1e-30f); }
inline float host_up(float v) { return v + (fabsf(v) * 0x1p-22f + 1e-30f); }
inline float bits_float(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct HostBox {
    float lo[3], hi[3];
    uint32_t low;
};

// the material records of every slot, in slot order, behind the pairs (rt_device.h BvhTables): from the finished index section
void host_fill_materials(float4 *blob, const std::vector<rt_sphere> &sph, uint32_t n_leaves, uint32_t n_slots) {
    const uint32_t *index = reinterpret_cast<const uint32_t *>(blob + rt::bvh_index_at(n_slots));
    float4 *emis = blob + rt::bvh_emis_at(n_leaves, n_slots), *colr = blob + rt::bvh_colr_at(n_leaves, n_slots);
    for (uint32_t j = 0; j < n_slots; ++j) {
        const uint32_t ix = index[j];
        if (ix == 0xffffffffu) {
            emis[j] = colr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const rt_sphere &q = sph[ix];
        float refl_bits;
        memcpy(&refl_bits, &q.refl, 4);
        emis[j] = make_float4(q.e.x, q.e.y, q.e.z, refl_bits);
        colr[j] = make_float4(q.c.x, q.c.y, q.c.z, q.rad);
    }
}

inline double host_box_area(const HostBox &b) {
    const double dx = (double)b.hi[0] - b.lo[0], dy = (double)b.hi[1] - b.lo[1], dz = (double)b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

int build_on_host(rt_ctx *c, uint32_t n_total, float r_cut, float r_floor, uint32_t n_always, uint32_t n_tree, hipStream_t stream) {
    const std::vector<rt_sphere> &sph = c->h_spheres;
    const uint32_t n_leaves = (n_tree + rt::kBvhLeaf - 1) / rt::kBvhLeaf;
    const uint32_t n_slots = n_always + rt::kBvhLeaf * n_leaves;
    const size_t total4 = rt::bvh_blob_float4s(n_leaves, n_slots);
    if (c->bvh_stage_cap < total4) {
        if (c->bvh_stage_used) HIP_TRY(hipEventSynchronize(c->bvh_stage_ev));
        if (c->h_bvh_stage) (void)hipHostFree(c->h_bvh_stage);
        c->h_bvh_stage = nullptr;
        c->bvh_stage_cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_bvh_stage), total4 * sizeof(float4), hipHostMallocDefault));
        c->bvh_stage_cap = total4;
    } else if (c->bvh_stage_used) {
        HIP_TRY(hipEventSynchronize(c->bvh_stage_ev));       // the last build's copy still reads the buffer
    }
    float4 *blob = c->h_bvh_stage;
    float4 *hdr = blob, *slots = blob + rt::bvh_slots_at();
    uint32_t *index = reinterpret_cast<uint32_t *>(blob + rt::bvh_index_at(n_slots));
    float4 *pairs = blob + rt::bvh_pairs_at(n_slots);
    // split; radius range
    std::vector<uint32_t> order;
    order.reserve(n_tree);
    float rmin = 3.4e38f, rmin_all = 3.4e38f, rmax = 0.f;
    uint32_t na = 0;
    for (uint32_t i = 0; i < n_total; ++i) {
        const rt_sphere &s = sph[i];
        if (c->have_dups && c->h_dup_stage[i]) continue;
        if (bvh_outside(s.rad, s.p.x, s.p.y, s.p.z, r_cut)) {
            slots[na] = make_float4(s.p.x, s.p.y, s.p.z, s.rad * s.rad);
            index[na] = i;
            na += 1;
        } else {
            order.push_back(i);
            rmin_all = fminf(rmin_all, fabsf(s.rad));
            if (fabsf(s.rad) >= r_floor) rmin = fminf(rmin, fabsf(s.rad));
            rmax = fmaxf(rmax, fabsf(s.rad));
        }
    }
    const bool have_regular = rmin < 3.4e38f;          // (bvh_half_width: the header's r_min is the smallest regular radius, smaller spheres' boxes grow by half of it)
    if (!have_regular) rmin = rmin_all;
    const float grow = have_regular ? 0.5f * rmin : 0.f;
    if (na != n_always || order.size() != n_tree) return rt::fail(RT_ERR_STATE, "hierarchy: the split changed under the build");
    // order: top-down, every node's spheres partitioned at the median along the longest axis of the box of their centres
    // (the left child takes the first half of the node's leaves; ties go by scene index), as the device build does
    {
        auto coord = [&](uint32_t ix, int axis) { const rt_sphere &s = sph[ix]; return axis == 0 ? s.p.x : (axis == 1 ? s.p.y : s.p.z); };
        struct Range { uint32_t a, b; };
        std::vector<Range> todo{ { 0, n_leaves } };
        while (!todo.empty()) {
            const Range rg = todo.back();
            todo.pop_back();
            if (rg.b - rg.a <= 1) continue;
            const size_t first = (size_t)rg.a * rt::kBvhLeaf, last = std::min((size_t)rg.b * rt::kBvhLeaf, (size_t)n_tree);
            float lo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, hi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
            for (size_t j = first; j < last; ++j)
                for (int a3 = 0; a3 < 3; ++a3) {
                    lo[a3] = fminf(lo[a3], coord(order[j], a3));
                    hi[a3] = fmaxf(hi[a3], coord(order[j], a3));
                }
            const float ext[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] };
            int axis = 0;
            if (ext[1] > ext[axis]) axis = 1;
            if (ext[2] > ext[axis]) axis = 2;
            const uint32_t mid = (rg.a + rg.b) / 2;
            const size_t cut = std::min((size_t)mid * rt::kBvhLeaf, last);
            std::nth_element(order.begin() + first, order.begin() + cut, order.begin() + last, [&](uint32_t x, uint32_t y) {
                const float cx = coord(x, axis), cy = coord(y, axis);
                return cx < cy || (cx == cy && x < y);
            });
            todo.push_back({ rg.a, mid });
            todo.push_back({ mid, rg.b });
        }
    }
    // records in leaf order, leaf boxes
    std::vector<HostBox> leaf(n_leaves);
    const float qnan = bits_float(0x7fc00000u);
    for (uint32_t l = 0; l < n_leaves; ++l) {
        HostBox b{ { 3.4e38f, 3.4e38f, 3.4e38f }, { -3.4e38f, -3.4e38f, -3.4e38f }, 0xffffffffu };
        for (int q = 0; q < rt::kBvhLeaf; ++q) {
            const uint32_t j = rt::kBvhLeaf * l + q;
            if (j >= n_tree) {
                slots[n_always + j] = make_float4(qnan, qnan, qnan, qnan);
                index[n_always + j] = 0xffffffffu;
                continue;
            }
            const uint32_t ix = order[j];
            const rt_sphere &s = sph[ix];
            slots[n_always + j] = make_float4(s.p.x, s.p.y, s.p.z, s.rad * s.rad);
            index[n_always + j] = ix;
            const float p[3] = { s.p.x, s.p.y, s.p.z }, ar = bvh_half_width(fabsf(s.rad), r_floor, grow);
            for (int a = 0; a < 3; ++a) {
                b.lo[a] = fminf(b.lo[a], host_down(p[a] - ar));
                b.hi[a] = fmaxf(b.hi[a], host_up(p[a] + ar));
            }
            b.low = ix < b.low ? ix : b.low;
        }
        leaf[l] = b;
    }
    // sibling pairs: a recursion over leaf ranges (a range's box is the union of its halves')
    double area_inner = 0.0, area_leaf = 0.0;       // surface areas of the inner nodes below the root / of the leaves (estimate_forms)
    auto range_box = [&](uint32_t a, uint32_t b, auto &&self) -> HostBox {
        if (b - a == 1) return leaf[a];
        const uint32_t mid = (a + b) / 2;
        const HostBox L = self(a, mid, self), R = self(mid, b, self);
        const HostBox side[2] = { L, R };
        const uint32_t ca[2] = { a, mid }, cb[2] = { mid, b };
        for (int sd = 0; sd < 2; ++sd) {
            (cb[sd] - ca[sd] == 1 ? area_leaf : area_inner) += host_box_area(side[sd]);
            const uint32_t ref = (cb[sd] - ca[sd] == 1) ? (rt::kBvhLeafRef | ca[sd]) : (ca[sd] + cb[sd]) / 2 - 1;
            pairs[4 * (size_t)(mid - 1) + 2 * sd] = make_float4(side[sd].lo[0], side[sd].lo[1], side[sd].lo[2], bits_float(ref));
            pairs[4 * (size_t)(mid - 1) + 2 * sd + 1] = make_float4(side[sd].hi[0], side[sd].hi[1], side[sd].hi[2], bits_float(side[sd].low));
        }
        HostBox u = L;
        for (int a3 = 0; a3 < 3; ++a3) {
            u.lo[a3] = fminf(L.lo[a3], R.lo[a3]);
            u.hi[a3] = fmaxf(L.hi[a3], R.hi[a3]);
        }
        u.low = L.low < R.low ? L.low : R.low;
        return u;
    };
    const HostBox root = range_box(0, n_leaves, range_box);     // (depth = log2 of the leaf count: 15 at most)
    const float cx = 0.5f * root.lo[0] + 0.5f * root.hi[0], cy = 0.5f * root.lo[1] + 0.5f * root.hi[1], cz = 0.5f * root.lo[2] + 0.5f * root.hi[2];
    const float ex = root.hi[0] - cx, ey = root.hi[1] - cy, ez = root.hi[2] - cz;
    hdr[0] = make_float4(cx, cy, cz, sqrtf(ex * ex + ey * ey + ez * ez) * 1.001f);
    hdr[1] = make_float4(rmin, rmax, 1.f / (2.f * rmin), bits_float(n_leaves > 1 ? n_leaves / 2u - 1u : rt::kBvhLeafRef));
    host_fill_materials(blob, sph, n_leaves, n_slots);
    HIP_TRY(hipMemcpyAsync(c->d_bvh, blob, total4 * sizeof(float4), hipMemcpyHostToDevice, stream));
    if (!c->bvh_stage_ev) HIP_TRY(hipEventCreate(&c->bvh_stage_ev));
    HIP_TRY(hipEventRecord(c->bvh_stage_ev, stream));
    c->bvh_stage_used = true;
    const double a_root = host_box_area(root);
    if (n_leaves == 1) area_leaf = a_root;
    c->bvh_est_valid = a_root > 0.0 && std::isfinite(a_root);
    c->bvh_est_pairs = c->bvh_est_valid ? (n_leaves > 1 ? 1.0 : 0.0) + area_inner / a_root : 0.0;
    c->bvh_est_leaves = c->bvh_est_valid ? area_leaf / a_root : 0.0;
    return RT_OK;
}


// The same tables with the tree's SHAPE chosen by surface area (a full scene upload, where the host has the records and the call
// blocks anyway; device-resident updates keep the device build above and its fixed shape).  Top-down: a node's spheres are
// sorted along each axis in turn and cut where  area(left) * leaves(left) + area(right) * leaves(right)  is smallest
// (leaves(n) = ceil(n / 8): the cost of a visit is a leaf's eight sphere tests whether the leaf is full or not, so partial
// leaves are made only where they pay); 8 spheres or fewer are a leaf.  Leaves are numbered in the order the recursion emits
// them, every subtree holds a contiguous range of them, and the pair of a node sits at (first leaf of its right child) - 1 --
// the numbering of the fixed shape, which never depended on where the split lies.  The root's pair goes out through
// BvhTables::root since it is no longer n_leaves / 2 - 1.  Against the fixed shape, on C3's rays (a host model, profiles/r03y_tree_shape_model.txt): pair
// steps per ray -15 % (shadow rays -28 %), leaf visits -7 %.  Returns RT_OK with *built = false when the result does not fit
// the tables' allocation or the stack budget (the caller then takes the fixed shape).
constexpr uint32_t kSahMaxTree = rt::kAlwaysWalkFrom - 1;   // the host shapes the trees whose surface areas the choice of form is estimated from (1.2 ms at 1024
                                                // spheres); from kAlwaysWalkFrom on nothing is estimated and the device shapes the tree (rt_bvh_build_sah_kernel)
constexpr uint32_t kSahMinTree = 128;           // below 16 leaves the halved shape is as good (64 spheres: 6.95 against 7.03 ms) and one level shallower
constexpr uint32_t kSahMaxDepth = 30;
struct SahOut {
    HostBox box;
    uint32_t ref;
    uint32_t depth;
};
int build_on_host_sah(rt_ctx *c, uint32_t n_total, float r_cut, float r_floor, uint32_t n_always, uint32_t n_tree, hipStream_t stream, uint32_t *n_leaves_out,
                      uint32_t *depth_out, bool *built) {
    *built = false;
    const std::vector<rt_sphere> &sph = c->h_spheres;
    std::vector<uint32_t> order, always;
    order.reserve(n_tree);
    float rmin = 3.4e38f, rmin_all = 3.4e38f, rmax = 0.f;
    for (uint32_t i = 0; i < n_total; ++i) {
        const rt_sphere &s = sph[i];
        if (c->have_dups && c->h_dup_stage[i]) continue;
        if (bvh_outside(s.rad, s.p.x, s.p.y, s.p.z, r_cut)) {
            always.push_back(i);
        } else {
            order.push_back(i);
            rmin_all = fminf(rmin_all, fabsf(s.rad));
            if (fabsf(s.rad) >= r_floor) rmin = fminf(rmin, fabsf(s.rad));
            rmax = fmaxf(rmax, fabsf(s.rad));
        }
    }
    const bool have_regular = rmin < 3.4e38f;          // (bvh_half_width)
    if (!have_regular) rmin = rmin_all;
    const float grow_small = have_regular ? 0.5f * rmin : 0.f;
    if (always.size() != n_always || order.size() != n_tree) return rt::fail(RT_ERR_STATE, "hierarchy: the split changed under the build");
    auto coord = [&](uint32_t ix, int axis) { const rt_sphere &s = sph[ix]; return axis == 0 ? s.p.x : (axis == 1 ? s.p.y : s.p.z); };
    auto grow = [&](HostBox &b, uint32_t ix) {
        const rt_sphere &s = sph[ix];
        const float p[3] = { s.p.x, s.p.y, s.p.z }, ar = bvh_half_width(fabsf(s.rad), r_floor, grow_small);
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = fminf(b.lo[a], host_down(p[a] - ar));
            b.hi[a] = fmaxf(b.hi[a], host_up(p[a] + ar));
        }
        b.low = ix < b.low ? ix : b.low;
    };
    auto area = [](const HostBox &b) {
        const double dx = (double)b.hi[0] - b.lo[0], dy = (double)b.hi[1] - b.lo[1], dz = (double)b.hi[2] - b.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    const HostBox empty{ { 3.4e38f, 3.4e38f, 3.4e38f }, { -3.4e38f, -3.4e38f, -3.4e38f }, 0xffffffffu };
    std::vector<uint32_t> leaf_first, leaf_count;
    std::vector<float4> pair_rows;          // 4 per pair, at 4 * (mid - 1); grown as leaves are emitted
    std::vector<double> right_area;
    bool too_deep = false;
    double area_inner = 0.0, area_leaf = 0.0;       // as in build_on_host
    auto put_pair = [&](uint32_t mid, const SahOut &L, const SahOut &R) {
        if (pair_rows.size() < 4 * (size_t)mid) pair_rows.resize(4 * (size_t)mid, make_float4(0.f, 0.f, 0.f, 0.f));
        const SahOut *side[2] = { &L, &R };
        for (int sd = 0; sd < 2; ++sd) {
            ((side[sd]->ref & rt::kBvhLeafRef) ? area_leaf : area_inner) += area(side[sd]->box);
            pair_rows[4 * (size_t)(mid - 1) + 2 * sd] = make_float4(side[sd]->box.lo[0], side[sd]->box.lo[1], side[sd]->box.lo[2], bits_float(side[sd]->ref));
            pair_rows[4 * (size_t)(mid - 1) + 2 * sd + 1] = make_float4(side[sd]->box.hi[0], side[sd]->box.hi[1], side[sd]->box.hi[2], bits_float(side[sd]->box.low));
        }
    };
    auto by_axis = [&](int axis) {
        return [&coord, axis](uint32_t x, uint32_t y) {
            const float cx = coord(x, axis), cy = coord(y, axis);
            return cx < cy || (cx == cy && x < y);
        };
    };
    auto build = [&](size_t first, size_t last, uint32_t depth, auto &&self) -> SahOut {
        const size_t count = last - first;
        if (depth > kSahMaxDepth) too_deep = true;
        if (count <= (size_t)rt::kBvhLeaf || too_deep) {
            // (too deep: the rest becomes leaves of 8 in whatever order it is in -- the result is discarded anyway)
            SahOut out{ empty, 0u, 1u };
            if (count <= (size_t)rt::kBvhLeaf) {
                const uint32_t id = (uint32_t)leaf_first.size();
                leaf_first.push_back((uint32_t)first);
                leaf_count.push_back((uint32_t)count);
                for (size_t j = first; j < last; ++j) grow(out.box, order[j]);
                out.ref = rt::kBvhLeafRef | id;
                return out;
            }
            const size_t half = first + ((count / 2 + rt::kBvhLeaf - 1) / rt::kBvhLeaf) * rt::kBvhLeaf;
            const SahOut L = self(first, std::min(half, last - 1), depth + 1, self);
            const uint32_t mid = (uint32_t)leaf_first.size();
            const SahOut R = self(std::min(half, last - 1), last, depth + 1, self);
            put_pair(mid, L, R);
            out.box = L.box;
            for (int a = 0; a < 3; ++a) { out.box.lo[a] = fminf(L.box.lo[a], R.box.lo[a]); out.box.hi[a] = fmaxf(L.box.hi[a], R.box.hi[a]); }
            out.box.low = L.box.low < R.box.low ? L.box.low : R.box.low;
            out.ref = mid - 1u;
            out.depth = 1u + (L.depth > R.depth ? L.depth : R.depth);
            return out;
        }
        int best_axis = 0;
        size_t best_cut = count / 2;
        double best = 1e300;
        right_area.resize(count);
        for (int axis = 0; axis < 3; ++axis) {
            std::sort(order.begin() + first, order.begin() + last, by_axis(axis));
            HostBox b = empty;
            for (size_t i = count; i-- > 1;) {              // right_area[i] = area of spheres [i, count)
                grow(b, order[first + i]);
                right_area[i] = area(b);
            }
            b = empty;
            for (size_t cut = 1; cut < count; ++cut) {
                grow(b, order[first + cut - 1]);
                const double cost = area(b) * (double)((cut + rt::kBvhLeaf - 1) / rt::kBvhLeaf) +
                                    right_area[cut] * (double)((count - cut + rt::kBvhLeaf - 1) / rt::kBvhLeaf);
                if (cost < best) {
                    best = cost;
                    best_axis = axis;
                    best_cut = cut;
                }
            }
        }
        if (best_axis != 2) std::sort(order.begin() + first, order.begin() + last, by_axis(best_axis));
        const SahOut L = self(first, first + best_cut, depth + 1, self);
        const uint32_t mid = (uint32_t)leaf_first.size();
        const SahOut R = self(first + best_cut, last, depth + 1, self);
        put_pair(mid, L, R);
        SahOut out{ L.box, mid - 1u, 1u + (L.depth > R.depth ? L.depth : R.depth) };
        for (int a = 0; a < 3; ++a) { out.box.lo[a] = fminf(L.box.lo[a], R.box.lo[a]); out.box.hi[a] = fmaxf(L.box.hi[a], R.box.hi[a]); }
        out.box.low = L.box.low < R.box.low ? L.box.low : R.box.low;
        return out;
    };
    const SahOut root = build(0, order.size(), 1u, build);
    const uint32_t n_leaves = (uint32_t)leaf_first.size();
    const uint32_t n_slots = n_always + rt::kBvhLeaf * n_leaves;
    const size_t total4 = rt::bvh_blob_float4s(n_leaves, n_slots);
    if (too_deep || n_leaves >= rt::kBvhLeafRef || total4 > (size_t)c->scene_cap * 6 + 64) return RT_OK;      // (the allocation of ensure_scene_capacity)
    if (c->bvh_stage_cap < total4) {
        if (c->bvh_stage_used) HIP_TRY(hipEventSynchronize(c->bvh_stage_ev));
        if (c->h_bvh_stage) (void)hipHostFree(c->h_bvh_stage);
        c->h_bvh_stage = nullptr;
        c->bvh_stage_cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_bvh_stage), total4 * sizeof(float4), hipHostMallocDefault));
        c->bvh_stage_cap = total4;
    } else if (c->bvh_stage_used) {
        HIP_TRY(hipEventSynchronize(c->bvh_stage_ev));       // the last build's copy still reads the buffer
    }
    float4 *blob = c->h_bvh_stage;
    float4 *hdr = blob, *slots = blob + rt::bvh_slots_at();
    uint32_t *index = reinterpret_cast<uint32_t *>(blob + rt::bvh_index_at(n_slots));
    float4 *pairs = blob + rt::bvh_pairs_at(n_slots);
    memset(blob, 0, total4 * sizeof(float4));
    for (uint32_t k = 0; k < n_always; ++k) {
        const rt_sphere &s = sph[always[k]];
        slots[k] = make_float4(s.p.x, s.p.y, s.p.z, s.rad * s.rad);
        index[k] = always[k];
    }
    const float qnan = bits_float(0x7fc00000u);
    for (uint32_t l = 0; l < n_leaves; ++l)
        for (int q = 0; q < rt::kBvhLeaf; ++q) {
            const size_t at = (size_t)n_always + (size_t)rt::kBvhLeaf * l + q;
            if ((uint32_t)q >= leaf_count[l]) {
                slots[at] = make_float4(qnan, qnan, qnan, qnan);
                index[at] = 0xffffffffu;
                continue;
            }
            const uint32_t ix = order[leaf_first[l] + q];
            const rt_sphere &s = sph[ix];
            slots[at] = make_float4(s.p.x, s.p.y, s.p.z, s.rad * s.rad);
            index[at] = ix;
        }
    for (size_t k = 0; k < 4 * (size_t)(n_leaves ? n_leaves - 1 : 0) && k < pair_rows.size(); ++k) pairs[k] = pair_rows[k];
    const HostBox &rb = root.box;
    const float cx = 0.5f * rb.lo[0] + 0.5f * rb.hi[0], cy = 0.5f * rb.lo[1] + 0.5f * rb.hi[1], cz = 0.5f * rb.lo[2] + 0.5f * rb.hi[2];
    const float ex = rb.hi[0] - cx, ey = rb.hi[1] - cy, ez = rb.hi[2] - cz;
    hdr[0] = make_float4(cx, cy, cz, sqrtf(ex * ex + ey * ey + ez * ez) * 1.001f);
    hdr[1] = make_float4(rmin, rmax, 1.f / (2.f * rmin), bits_float(n_leaves > 1 ? root.ref : rt::kBvhLeafRef));
    host_fill_materials(blob, sph, n_leaves, n_slots);
    HIP_TRY(hipMemcpyAsync(c->d_bvh, blob, total4 * sizeof(float4), hipMemcpyHostToDevice, stream));
    if (!c->bvh_stage_ev) HIP_TRY(hipEventCreate(&c->bvh_stage_ev));
    HIP_TRY(hipEventRecord(c->bvh_stage_ev, stream));
    c->bvh_stage_used = true;
    *n_leaves_out = n_leaves;
    *depth_out = root.depth;
    *built = true;
    const double a_root = area(rb);
    if (n_leaves == 1) area_leaf = a_root;
    c->bvh_est_valid = a_root > 0.0 && std::isfinite(a_root);
    c->bvh_est_pairs = c->bvh_est_valid ? (n_leaves > 1 ? 1.0 : 0.0) + area_inner / a_root : 0.0;
    c->bvh_est_leaves = c->bvh_est_valid ? area_leaf / a_root : 0.0;
    return RT_OK;
}

}  // namespace

namespace rt {

hipError_t prepare_bvh_build() {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rt_bvh_build_sah_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(rt_bvh_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
}

// The hierarchy of a large scene (rt_bvh_build_kernel), on `stream` behind the records.  Which spheres stay outside
// the tree is decided here, from the host mirror, with the test the device applies to the same bits: the cut is 16 times the
// median |radius| and at least an eighth of the scene's extent (ground, walls, lights of the scene's size), non-finite records stay
// outside as well.
// After any builder: the top of the tree to the front of the pair table, where the walk will read its tables from HBM / L2 -- i.e. where not even
// the pairs fit the LDS budget the context gives the hierarchy (rt_launch.hip makes the same comparison per launch).  Costs one small launch per
// build (20-60 us); trees that are walked from LDS keep the builders' numbering.
static int promote_top(rt_ctx *c, hipStream_t stream) {
    c->bvh.n_top = 0;
#if !RT_DIAGNOSTICS
    (void)stream;
    return RT_OK;           // (the arm that stages the promoted top lost its A/B: the product library promotes nothing)
#else
    if (!c->bvh_ok || c->bvh.n_leaves < 2 || c->bvh_top_pairs <= 0) return RT_OK;
    const uint32_t n_pairs = c->bvh.n_leaves - 1u;
    const size_t pairs_lds = rt::lds_bytes_pairs(0, 0, false, 64, c->bvh.n_leaves, 0, c->bvh.stack_depth, 256);
    if (!RT_DIAGNOSTICS && c->bvh_mixed != 0 && pairs_lds <= (size_t)c->bvh_lds_limit) return RT_OK;       // (rt_trace_*_pairs / _pairs_m will walk it: everything it chases is in LDS; the diagnostics library, where any instance may be asked for by name, always promotes)
    const size_t used = rt::bvh_blob_float4s(c->bvh.n_leaves, c->bvh.n_slots);
    const size_t scratch4 = 4 * (size_t)n_pairs + ((size_t)n_pairs * 2 + 15) / 16 + 1;
    if (used + scratch4 > (size_t)c->scene_cap * 6 + 64) return RT_OK;                   // (no room behind the blob: the tree stays as it is, nothing is staged)
    uint32_t top = std::min<uint32_t>((uint32_t)c->bvh_top_pairs, rt::kBvhTopPairs);
#if RT_DIAGNOSTICS
    if (const char *e = getenv("RT_TOP_PAIRS")) top = std::min<uint32_t>((uint32_t)atoi(e), 256u);       // (experiments: how much of the top is worth staging)
    if (top == 0u) return RT_OK;
#endif
    hipLaunchKernelGGL(rt_bvh_promote_kernel, dim3(1), dim3(1024), 0, stream, c->d_bvh, c->bvh.n_slots, c->bvh.n_leaves, top, c->d_bvh + used);
    HIP_TRY(hipGetLastError());
    c->bvh.n_top = std::min(top, n_pairs);
    return RT_OK;
#endif
}

// ... and the packed pair table behind the blob (where the promotion's scratch was: it is dead by then -- same stream), for the same trees
static int pack_pairs(rt_ctx *c, hipStream_t stream) {
    c->bvh.packed_at = 0;
#if !RT_DIAGNOSTICS
    (void)stream;
    return RT_OK;           // (likewise: no product kernel reads a packed table)
#else
    if (!c->bvh_ok || c->bvh.n_leaves < 2 || c->bvh_packed == 0) return RT_OK;
    const uint32_t n_pairs = c->bvh.n_leaves - 1u;
    const size_t pairs_lds = rt::lds_bytes_pairs(0, 0, false, 64, c->bvh.n_leaves, 0, c->bvh.stack_depth, 256);
    if (!RT_DIAGNOSTICS && c->bvh_mixed != 0 && pairs_lds <= (size_t)c->bvh_lds_limit) return RT_OK;       // (walked from LDS: nothing reads a packed table -- but for an instance asked for by name in the diagnostics library)
    const size_t used = rt::bvh_blob_float4s(c->bvh.n_leaves, c->bvh.n_slots);
    if (used + 2 + 2 * (size_t)n_pairs > (size_t)c->scene_cap * 6 + 64) return RT_OK;
    const unsigned blocks = (unsigned)std::min<size_t>((n_pairs + 255) / 256, 1024);
    hipLaunchKernelGGL(rt_bvh_pack_pairs_kernel, dim3(blocks), dim3(256), 0, stream, c->d_bvh, c->bvh.n_slots, c->bvh.n_leaves, c->d_bvh + used);
    HIP_TRY(hipGetLastError());
    c->bvh.packed_at = (uint32_t)used;
    return RT_OK;
#endif
}

// Records that repeat an EARLIER record bit for bit in everything a ray test reads -- centre and radius^2 (SceneTables geom) -- can never be
// the answer to a ray: the test returns the same distance for both, and the reference's loops keep the first of equals (a closest hit
// takes a strictly nearer one only, .cl:215-232; a shadow ray stops at the first blocker, .cl:234-247) -- the lowest scene index wins.  So
// they stay out of the hierarchy altogether, always-list included: nothing observable changes (the counters are formed from scene indices, not
// from what the walk executes), and what the reference's own loader does to every scene file -- N zero-radius records at the origin in
// front of the N real ones, Utility.cpp:120,154 -- stops costing complex.scn's walk 98 leaves stacked on one point.  One byte per record, from
// the host mirror (open addressing over the four words; 3 ns per record), uploaded through page-locked staging only while the scene has
// such records.  Materials play no part: a repeated record's material is never read.
static int mark_duplicates(rt_ctx *c, uint32_t n_total, hipStream_t stream) {
    const bool had = c->have_dups;
    c->have_dups = false;
    c->n_dups = 0;
    if (n_total < 2 || !c->d_dup) return RT_OK;
    uint32_t cap = 16;
    while (cap < 2u * n_total) cap *= 2;
    static thread_local std::vector<uint32_t> table;
    table.assign(cap, 0u);                      // record index + 1
    static thread_local std::vector<uint8_t> flags;
    flags.assign(n_total, 0);
    auto key_of = [&](uint32_t i, uint32_t k[4]) {
        const rt_sphere &s = c->h_spheres[i];
        const float rr = s.rad * s.rad;
        memcpy(&k[0], &s.p.x, 4); memcpy(&k[1], &s.p.y, 4); memcpy(&k[2], &s.p.z, 4); memcpy(&k[3], &rr, 4);
    };
    uint32_t found = 0;
    for (uint32_t i = 0; i < n_total; ++i) {
        uint32_t k[4];
        key_of(i, k);
        const rt_sphere &s = c->h_spheres[i];
        if (!(fabsf(s.rad) <= 3.0e38f && fabsf(s.p.x) <= 3.0e38f && fabsf(s.p.y) <= 3.0e38f && fabsf(s.p.z) <= 3.0e38f)) continue;    // (NaN never equals itself; infinities stay as they are)
        uint32_t h = k[0] * 0x9E3779B1u ^ (k[1] + 0x7F4A7C15u) * 0x85EBCA77u ^ (k[2] + 0x165667B1u) * 0xC2B2AE3Du ^ (k[3] + 0x27D4EB2Fu) * 0x2545F491u;
        h ^= h >> 15;
        for (uint32_t at = h & (cap - 1);; at = (at + 1) & (cap - 1)) {
            const uint32_t e = table[at];
            if (e == 0u) { table[at] = i + 1u; break; }
            uint32_t q[4];
            key_of(e - 1u, q);
            if (q[0] == k[0] && q[1] == k[1] && q[2] == k[2] && q[3] == k[3]) { flags[i] = 1; found += 1; break; }
        }
    }
    if (found == 0 && !had) return RT_OK;       // (nothing to say, and the device holds no flags of an earlier scene)
    if (!c->h_dup_stage) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_dup_stage), (size_t)c->scene_cap, hipHostMallocDefault));
    if (c->dup_stage_used) HIP_TRY(hipEventSynchronize(c->dup_ev));
    memcpy(c->h_dup_stage, flags.data(), n_total);
    HIP_TRY(hipMemcpyAsync(c->d_dup, c->h_dup_stage, n_total, hipMemcpyHostToDevice, stream));
    if (!c->dup_ev) HIP_TRY(hipEventCreate(&c->dup_ev));
    HIP_TRY(hipEventRecord(c->dup_ev, stream));
    c->dup_stage_used = true;
    c->have_dups = found != 0;
    c->n_dups = found;
    return RT_OK;
}

static int build_bvh_tables(rt_ctx *c, uint32_t n_total, hipStream_t stream, bool full_upload);
int build_bvh(rt_ctx *c, uint32_t n_total, hipStream_t stream, bool full_upload) {
    int rc = build_bvh_tables(c, n_total, stream, full_upload);
    if (rc == RT_OK) rc = promote_top(c, stream);
    return rc != RT_OK ? rc : pack_pairs(c, stream);
}

static int build_bvh_tables(rt_ctx *c, uint32_t n_total, hipStream_t stream, bool full_upload) {
    c->bvh_ok = false;
    c->bvh = rt::BvhTables{};
    c->bvh_n_tree = 0;
    if (full_upload) c->bvh_est_valid = false;          // (a device-resident update keeps the upload's estimate: the probe logic decides when it is stale)
    if (c->bvh_min <= 0 || n_total < (uint32_t)c->bvh_min || !c->d_bvh) return RT_OK;
    {
        const int rc = mark_duplicates(c, n_total, stream);
        if (rc != RT_OK) return rc;
    }
    const uint8_t *d_dup = c->have_dups ? c->d_dup : nullptr;
    auto repeated = [&](uint32_t i) { return c->have_dups && c->h_dup_stage[i] != 0; };
    std::vector<float> radii;
    radii.reserve(n_total);
    for (uint32_t i = 0; i < n_total; ++i) {
        const float r = fabsf(c->h_spheres[i].rad);
        if (r <= 3.0e38f && r > 0.f) radii.push_back(r);       // (zero-radius records -- the loader's phantoms -- say nothing about the scene's scale)
    }
    if (radii.empty()) return RT_OK;
    std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
    const float r_median = radii[radii.size() / 2];
    const float r_floor = r_median / 16.f;                         // radii below this are "small": bvh_half_width
    // The cut: a sphere stays outside the tree when it is of the SCENE's size -- a ground plane, a wall: its box would lie over every box above it, every
    // ray visits it anyway.  16 x the median radius says that for scenes of one size class (every BASELINE and reference scene: their cut is this term).
    // A scene of two classes -- thousands of small spheres ("dust") among hundreds of objects fifty times their size -- put every object outside by that
    // term alone, and every ray swept them all: 6 000 small + 4 000 large spheres 27 ms a pass at 1080p against 0.5 ms for the small ones alone
    // (profiles/r06_always_list.jsonl).  So the cut is never below an eighth of the extent of the scene itself: the 2 % .. 98 % range of the centres of the
    // spheres under the first term, along the widest axis (quantiles: one record far away does not stretch it).  Nothing else depends on it: the
    // builders take the radius range for the walk's pad from what is IN the tree.
    float r_cut = 16.f * r_median;
    // (asked only when the first term would put MORE THAN 8 spheres outside: a ground plane, six walls, a light are swept at no cost worth a hierarchy,
    // and every scene that has no more than those -- every BASELINE and reference scene -- pays nothing for the question.  Not more than 8: in a scene of
    // 150 spheres with radii over three decades, 15 above the first term cost the walk as much as its tree -- 0.50 against 0.20 ms, profiles/r06_choice_fuzz.jsonl)
    uint32_t n_over = 0;
    for (uint32_t i = 0; i < n_total && n_over <= 8; ++i) {
        const rt_sphere &s = c->h_spheres[i];
        const float ar = fabsf(s.rad);
        n_over += (ar > r_cut && ar <= 3.0e38f && fabsf(s.p.x) <= 3.0e38f && fabsf(s.p.y) <= 3.0e38f && fabsf(s.p.z) <= 3.0e38f && !repeated(i)) ? 1u : 0u;
    }
    if (n_over > 8) {
        std::vector<float> axis[3];
        for (uint32_t i = 0; i < n_total; ++i) {
            const rt_sphere &s = c->h_spheres[i];
            if (repeated(i) || bvh_outside(s.rad, s.p.x, s.p.y, s.p.z, r_cut) || !(fabsf(s.rad) > 0.f)) continue;
            axis[0].push_back(s.p.x);
            axis[1].push_back(s.p.y);
            axis[2].push_back(s.p.z);
        }
        float extent = 0.f;
        if (axis[0].size() >= 50)
            for (int a = 0; a < 3; ++a) {
                const size_t m = axis[a].size(), lo = m / 50, hi = m - 1 - m / 50;
                std::nth_element(axis[a].begin(), axis[a].begin() + lo, axis[a].end());
                const float q_lo = axis[a][lo];
                std::nth_element(axis[a].begin(), axis[a].begin() + hi, axis[a].end());
                extent = std::max(extent, axis[a][hi] - q_lo);
            }
        if (extent <= 3.0e38f) r_cut = std::max(r_cut, extent / 8.f);
    }
    uint32_t n_tree = 0;
    for (uint32_t i = 0; i < n_total; ++i) {
        const rt_sphere &s = c->h_spheres[i];
        n_tree += (repeated(i) || bvh_outside(s.rad, s.p.x, s.p.y, s.p.z, r_cut)) ? 0u : 1u;
    }
    if (n_tree < (uint32_t)c->bvh_min) return RT_OK;
    const uint32_t n_always = n_total - c->n_dups - n_tree;
    const uint32_t n_leaves = (n_tree + rt::kBvhLeaf - 1) / rt::kBvhLeaf;
    c->bvh_n_tree = n_tree;
    // a full upload (rt_set_scene: the call blocks and the host has every record): the shape by surface area, on the host
    if (full_upload && c->bvh_sah == 1 && n_tree >= kSahMinTree && n_tree <= kSahMaxTree) {
        uint32_t sah_leaves = 0, sah_depth = 0;
        bool built = false;
        const int rc = build_on_host_sah(c, n_total, r_cut, r_floor, n_always, n_tree, stream, &sah_leaves, &sah_depth, &built);
        if (rc != RT_OK) return rc;
        if (built) {
            c->bvh = rt::BvhTables{ c->d_bvh, n_always, sah_leaves, n_always + rt::kBvhLeaf * sah_leaves, sah_depth,
                                    rt::bvh_emis_at(sah_leaves, n_always + rt::kBvhLeaf * sah_leaves) };
            c->bvh_ok = true;
            return RT_OK;
        }
    }
    // a full upload of a small tree: the fixed shape built on the host (the same tree the device builds: same splits, same
    // boxes), because the host then knows the surface areas the choice between hierarchy and sweep is estimated from
    const bool host_small = full_upload && n_tree < kSahMinTree;
    // device-resident updates, and uploads too large for the host to shape in passing: the shape by surface area, on the device
    if (c->bvh_sah && n_tree >= kSahMinTree && n_tree <= kSahDeviceMaxTree) {
        uint32_t n_pad = 2;
        while (n_pad < n_tree) n_pad *= 2;
        uint32_t depth_cap = 1;
        while ((1u << depth_cap) < n_leaves) depth_cap += 1;
        uint32_t margin = 2;                                    // (levels more than the halved shape needs: room for uneven cuts; 1, 2 and 3 give C3 the same tree, profiles/r04p_*)
#if RT_DIAGNOSTICS
        if (const char *e = getenv("RT_SAH_DEPTH_MARGIN")) margin = (uint32_t)atoi(e);
#endif
        depth_cap += margin;
        const size_t lds = (size_t)n_pad * 8 + (size_t)n_leaves * 74;
        hipLaunchKernelGGL(rt_bvh_build_sah_kernel, dim3(1), dim3(1024), lds, stream, c->d_spheres, d_dup, n_total, r_cut, r_floor, n_always, n_tree, n_pad, depth_cap, c->d_bvh);
        HIP_TRY(hipGetLastError());
        c->bvh = rt::BvhTables{ c->d_bvh, n_always, n_leaves, n_always + rt::kBvhLeaf * n_leaves, depth_cap + 1, rt::bvh_emis_at(n_leaves, n_always + rt::kBvhLeaf * n_leaves) };
        c->bvh_ok = true;
        return RT_OK;
    }
    if (n_tree <= kDeviceBuildMax && !host_small) {
        uint32_t n_pad = 2;
        while (n_pad < n_tree) n_pad *= 2;
        uint32_t level_nodes = 1;
        while (level_nodes < n_leaves) level_nodes *= 2;
        const size_t lds = std::max((size_t)n_pad * 8 + (size_t)level_nodes * 28, (size_t)n_leaves * 32);
        hipLaunchKernelGGL(rt_bvh_build_kernel, dim3(1), dim3(1024), lds, stream, c->d_spheres, d_dup, n_total, r_cut, r_floor, n_always, n_tree, n_pad, c->d_bvh);
        HIP_TRY(hipGetLastError());
    } else {
        const int rc = build_on_host(c, n_total, r_cut, r_floor, n_always, n_tree, stream);
        if (rc != RT_OK) return rc;
    }
    uint32_t depth = 1;
    while ((1u << depth) < n_leaves) depth += 1;
    c->bvh = rt::BvhTables{ c->d_bvh, n_always, n_leaves, n_always + rt::kBvhLeaf * n_leaves, depth + 1, rt::bvh_emis_at(n_leaves, n_always + rt::kBvhLeaf * n_leaves) };
    c->bvh_ok = true;
    return RT_OK;
}

}  // namespace rt
