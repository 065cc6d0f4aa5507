// rt_trace.inc.h -- the per-pixel path-trace kernel, included once per arithmetic mode.
//
//   RT_FAST = 0  (rt_kernel_parity.hip, -ffp-contract=off): every product, sum, quotient and
//                square root is a separately rounded binary32 operation in the association
//                order of the reference expression (RayTracing_Kernel.cl, cited ".cl:LINE");
//                sin/cos/pow are the restated-libm set of rt_detmath.h.  Output is bit-equal
//                to the reference kernel compiled as host C++.
//   RT_FAST = 1  (rt_kernel_fast.hip, -ffp-contract=fast): same algorithm, fused multiply-
//                adds, v_rcp/v_rsq/v_sqrt/v_sin/v_cos/v_exp/v_log.
//
// Mapping: one wavefront lane = one pixel; a 256-thread workgroup covers a 32x8 pixel tile,
// each of its 4 wavefronts an 8x8 sub-tile (coherent primary rays, 32-byte store segments).
// The sphere geometry table {centre, radius^2} is staged into LDS once per workgroup and read
// back as wave-uniform (broadcast) ds_read_b128 in the closest-hit and any-hit loops.
// The whole spp loop runs inside the launch: the seed pair, the running-average colour and
// the path state stay in registers; HBM sees one seed read, one seed write, one colour write
// and one packed-pixel write per pixel per launch (32 B/pixel).
//
// Lane-level path regeneration: lanes do not wait for each other at sample boundaries.  A
// lane whose path ends accumulates its sample and starts its next camera ray in the very next
// trip of the loop, so a wavefront only idles at the tail of the spp loop.  Each pixel still
// consumes its own RNG stream in the reference order, so results do not depend on this.
#include <hip/hip_runtime.h>

#include "rt_detmath.h"
#include "rt_device.h"

#ifndef RT_FAST
#error "define RT_FAST to 0 or 1"
#endif
#ifndef RT_DIAGNOSTICS
#define RT_DIAGNOSTICS 0
#endif
// Per-instance options (rt_kernel_parity.hip / rt_kernel_fast.hip define them before each include; rt_opts_reset.h
// forgets them).  The sweep shape itself is fixed: two spheres per loop trip (independent tests interleave, LDS reads
// are issued ahead of use) and the square-root half of a test skipped when NO lane of the wavefront has a non-negative
// discriminant (wave ballot).  What was measured against it and dropped is listed in DESIGN.md section 5 with the
// commits that hold the code.
//   RT_OPT_WG_WAVES       wavefronts per workgroup, 4 (a 32x8 tile) or 1 (an 8x8 tile).  A workgroup of four holds its
//                         wave slots until the dispatcher finds room for four more at once; single-wavefront workgroups
//                         refill slot by slot and give the heavy-first order an 8x8 granule.  Each workgroup stages its
//                         own copy of the tables, so 1 is for scenes whose tables are small (rt_api.hip picks).
//   RT_OPT_COOP           1: shadow rays of a wavefront share the idle lanes (coop_any; scenes of 12 spheres and more);
//                         2: verification instance, the sequential sweep runs beside it (diagnostics)
//   RT_OPT_WALK           1: large scenes -- the small spheres hang in a hierarchy that each lane walks for its own ray
//                         (rt_walk.inc.h, its own kernel body); 2: the same with a census of its steps (diagnostics)
//   RT_OPT_GLOBAL_TABLES  tables beyond the LDS budget are read where they lie in HBM / L2 (the plain sweep: through the scalar cache)
//   RT_OPT_MINWAVES       launch bound: wavefronts per SIMD the register allocation must allow
// Diagnostics build only:
//   RT_OPT_PERSIST        persistent wavefronts: the grid only fills the machine, each wavefront pulls 8x8 pixel tiles
//                         from a global queue and hands their pixels to its lanes one by one as lanes finish
//   RT_OPT_STAMPS         section census (executions and active lanes per section, counters[8..19])
//   RT_OPT_TIMELOG        device wall clock (s_memrealtime) of the launch and of every wavefront (P.timelog / P.wavelog)
#ifndef RT_OPT_WG_WAVES
#define RT_OPT_WG_WAVES 4
#endif
#ifndef RT_OPT_COOP
#define RT_OPT_COOP 0
#endif
#ifndef RT_OPT_WALK
#define RT_OPT_WALK 0
#endif
#ifndef RT_OPT_GLOBAL_TABLES
#define RT_OPT_GLOBAL_TABLES 0
#endif
#ifndef RT_OPT_MINWAVES
#define RT_OPT_MINWAVES 1
#endif
#ifndef RT_OPT_PERSIST
#define RT_OPT_PERSIST 0
#endif
// Fused mode, diagnostics build only (the experiment VERDICT r4 item 1c asked for, profiles/r05_fast_gate.jsonl): every DECISION of
// a path -- the sphere test's discriminant, root and epsilon compares (.cl:173-201), which side of a light and of the surface a
// sample lies on (.cl:283-296), total internal reflection (.cl:438), the Fresnel roulette (.cl:470) -- in uncontracted,
// correctly rounded binary32 as in parity mode; everything continuous (hit points, normals, directions, weights, sine / cosine,
// gamma) fused and on the hardware's approximations as in fast mode.
#ifndef RT_OPT_RAYS2
#define RT_OPT_RAYS2 0                  /* diagnostics: the hierarchy walk with two pixels per lane (rt_walk.inc.h) */
#endif
#ifndef RT_OPT_TOP_PAIRS
#define RT_OPT_TOP_PAIRS 0          /* with RT_OPT_WALK and RT_OPT_GLOBAL_TABLES 1: the promoted top of the tree (BvhTables::n_top pairs) staged in LDS */
#endif
#ifndef RT_OPT_PACKED_PAIRS
#define RT_OPT_PACKED_PAIRS 0       /* with RT_OPT_WALK and RT_OPT_GLOBAL_TABLES 1: the walk reads the packed pair table (32 bytes per pair, rt_device.h) */
#endif
#ifndef RT_OPT_PREFETCH
#define RT_OPT_PREFETCH 0           /* diagnostics, with RT_OPT_WALK and RT_OPT_GLOBAL_TABLES 1: both children's records requested one level ahead (rt_walk.inc.h) */
#endif
#ifndef RT_OPT_PAIR_PLANES
#define RT_OPT_PAIR_PLANES 0            /* diagnostics: the walk's staged pairs in four 16-byte planes (rt_walk.inc.h) */
#endif
#ifndef RT_OPT_EXACT_DECISIONS
#define RT_OPT_EXACT_DECISIONS 0
#endif
#ifndef RT_OPT_STAMPS
#define RT_OPT_STAMPS 0
#endif
#ifndef RT_OPT_TIMELOG
#define RT_OPT_TIMELOG 0
#endif
// Heavy tiles first (P.order / P.tile_cost): every workgroup leaves the wall-clock time of its slowest
// wavefront in P.tile_cost[tile]; a later launch of the same scene and camera walks the tiles in
// descending order of that cost (P.order), so that the launch ends on cheap tiles (sky) instead of on a
// few wavefronts of the most expensive ones (glass: up to 8 bounces x 64 samples, 1.7 ms against a mean of
// 0.5 ms) that happened to start late.  Scheduling only: pixels do not depend on who renders them when.


#undef RT_STAMP
#undef RT_STAMP_ROOTS
#if RT_OPT_STAMPS
// census: executions of each section per wavefront and active lanes in them.  The first active
// lane of the wavefront adds to a workgroup counter in LDS (a per-lane sum would only see the
// entries its own lane took part in).
#define RT_STAMP(k)                                                                          \
    do {                                                                                     \
        asm volatile("; @@SEC " #k);                                                          \
        const unsigned long long b_ = __builtin_amdgcn_ballot_w64(true);                     \
        if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)b_) - 1))                    \
            atomicAdd(&s_census[k], (1ull << 32) + (unsigned)__popcll(b_));                  \
    } while (0)
#define RT_STAMP_ROOTS(k, cnt)                                                               \
    do {                                                                                     \
        const unsigned long long b_ = __builtin_amdgcn_ballot_w64(true);                     \
        if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)b_) - 1))                    \
            atomicAdd(&s_census[k], (unsigned long long)(cnt));                              \
    } while (0)
#else
#define RT_STAMP(k)
#define RT_STAMP_ROOTS(k, cnt)
#endif

namespace rt {
namespace RT_NS {

#define RT_EPS 0.01f                    /* .cl:68 */
#define RT_PI 3.14159265358979323846f   /* .cl:69 */

struct V3 {
    float x, y, z;
};

RT_DEV V3 mk(float x, float y, float z) { return V3{ x, y, z }; }
RT_DEV V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 mul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 scale(V3 a, float k) { return mk(a.x * k, a.y * k, a.z * k); }
RT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // .cl:117-120
// A product in front of a decision: never contracted into the sum that follows where RT_OPT_EXACT_DECISIONS asks for that
// (parity mode never contracts).  Under -ffp-contract=fast the backend fuses whatever multiply and add it sees, whatever a
// pragma says, so the product is made opaque to it: an empty asm statement, no instruction.
RT_DEV float mul_decision(float a, float b) {
    float m = a * b;
#if RT_OPT_EXACT_DECISIONS && RT_FAST
    asm("" : "+v"(m));
#endif
    return m;
}
RT_DEV float dot_decision(V3 a, V3 b) { return mul_decision(a.x, b.x) + mul_decision(a.y, b.y) + mul_decision(a.z, b.z); }
RT_DEV V3 cross(V3 a, V3 b) {                                                        // .cl:128-131
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// Correctly rounded square root without the compiler's generic wrapper.  The compiler expands
// sqrtf into: scale tiny inputs by 2^32, v_sqrt_f32 (1 ulp), try the two neighbours with fused
// residuals, unscale, patch zero/infinity -- 16 VALU.  The scaling and the patch are only needed
// for nonzero inputs below 2^-96 in magnitude (the residuals stay normal otherwise; +-0, +inf, NaN and negative
// inputs come out of the neighbour test unchanged, as worked through in DESIGN.md), so one wave
// ballot routes the rare wavefront that holds such an input to the generic form and everybody
// else runs 9 VALU + the check.  Equal to sqrtf for all 2^32 inputs (rt_debug_sqrt_mismatches).
RT_DEV float ieee_sqrt_core(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float e_dn = __builtin_fmaf(-s_dn, s, x);
    const float e_up = __builtin_fmaf(-s_up, s, x);
    s = (e_dn <= 0.f) ? s_dn : s;
    s = (e_up > 0.f) ? s_up : s;
    return s;
}
RT_DEV float ieee_sqrt_lean(float x) {
    if (__builtin_amdgcn_ballot_w64(fabsf(x) < 0x1p-96f && x != 0.f) != 0ull) return sqrtf(x);
    return ieee_sqrt_core(x);
}

RT_DEV float rt_sqrt(float x) {
#if RT_FAST
    return __builtin_amdgcn_sqrtf(x);
#else
    return ieee_sqrt_lean(x);
#endif
}
// Square root of a value that is 0 or at least 2^-96 by construction (1 - z*z with z = 1 - 2k/2^23,
// k/2^23, 1 - k/2^23): no range check.
RT_DEV float rt_sqrt_unit(float x) {
#if RT_FAST
    return __builtin_amdgcn_sqrtf(x);
#else
    return ieee_sqrt_core(x);
#endif
}
// Square root of a discriminant (hit_post).  No range check either: for 0 < det < 2^-96 the
// unscaled form returns some sq with 0 <= sq < 2^-47 (never NaN), and the roots b -+ sq then do
// not depend on which: with |b| >= 2^-23 both round to b (sq is below half an ulp of b), and with
// |b| < 2^-23 both are below EPSILON and the test returns 0 (rt_debug_hitpost_mismatches tries
// every such det against a set of b on the device).
RT_DEV float rt_sqrt_det(float x) {
#if RT_FAST && !RT_OPT_EXACT_DECISIONS
    return __builtin_amdgcn_sqrtf(x);
#else
    return ieee_sqrt_core(x);
#endif
}
RT_DEV float rt_rcp(float x) {
#if RT_FAST
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.f / x;
#endif
}
RT_DEV float rt_div(float a, float b) {
#if RT_FAST
    return a * __builtin_amdgcn_rcpf(b);
#else
    return a / b;
#endif
}
// sqrt(dd) and 1/sqrt(dd), each correctly rounded (the reference's vnorm: f = 1.f/sqrt(v.v)).
// Lean form: v_rcp_f32 plus ONE fused Newton step equals the correctly rounded quotient 1.f/x for
// every x with biased exponent 1..252 (all 2^32 inputs tried on the device: tools/rcp_probe.py,
// rt_debug_rcp_probe), and for dd in [2^-96, FLT_MAX] the root lies in [2^-48, 2^64].  One wave
// ballot sends the rare wavefront holding anything else (zero, tiny, infinite, NaN) to the
// generic expansions.
RT_DEV float sqrt_and_rcp(float dd, float &root) {
#if RT_FAST
    root = __builtin_amdgcn_sqrtf(dd);
    return __builtin_amdgcn_rsqf(dd);
#else
    if (__builtin_amdgcn_ballot_w64(!(dd >= 0x1p-96f && dd <= 0x1.fffffep+127f)) != 0ull) {
        root = sqrtf(dd);
        return 1.f / root;
    }
    float s = __builtin_amdgcn_sqrtf(dd);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float e_dn = __builtin_fmaf(-s_dn, s, dd);
    const float e_up = __builtin_fmaf(-s_up, s, dd);
    s = (e_dn <= 0.f) ? s_dn : s;
    s = (e_up > 0.f) ? s_up : s;
    root = s;
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float e0 = __builtin_fmaf(-s, r0, 1.f);
    return __builtin_fmaf(e0, r0, r0);
#endif
}
RT_DEV V3 unit(V3 a) {                                                               // .cl:122-126
    float root;
    return scale(a, sqrt_and_rcp(dot(a, a), root));
}

// .cl:354 sign(): +-1, zero keeps its sign, NaN -> 0.  Three selects, no branches.
RT_DEV float cl_sign(float x) {
    float r = __builtin_copysignf(1.f, x);
    r = (x == 0.f) ? x : r;
    r = (x != x) ? 0.f : r;
    return r;
}

// .cl:143-169.  f lies in [2, 4) on a 2^-22 grid, so (f - 2) / 2 and f * 0.5 - 1 are both exact and
// equal: one fused operation (written out: this translation unit never contracts by itself).
// One multiply-with-carry step, a * (s & 65535) + (s >> 16): v_mad_u32_u16 multiplies the low 16
// bits of its operands itself, which saves the mask the compiler otherwise emits (it does not form
// this instruction on its own).
RT_DEV uint32_t mwc_step(uint32_t s, uint32_t a) {
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(s), "s"(a), "v"(s >> 16));
    return r;
}
RT_DEV uint32_t next_random_word(uint32_t &s0, uint32_t &s1) {
    s0 = mwc_step(s0, 36969u);
    s1 = mwc_step(s1, 18000u);
    uint32_t word = (s0 << 16) + s1;
    return (word & 0x007fffffu) | 0x40000000u;
}
RT_DEV float next_random(uint32_t &s0, uint32_t &s1) {
    return __builtin_fmaf(__uint_as_float(next_random_word(s0, s1)), 0.5f, -1.0f);
}
// GetRandom() - 0.5f (.cl:507-508): u lies on a 2^-23 grid in [0, 1), so u - 0.5 is exact too
RT_DEV float next_random_centred(uint32_t &s0, uint32_t &s1) {
    return __builtin_fmaf(__uint_as_float(next_random_word(s0, s1)), 0.5f, -1.5f);
}
// 1.f - 2.f * GetRandom() (.cl:204): 2u and 1 - 2u are exact (2^-22 grid in (-1, 1]) and equal 3 - f
RT_DEV float next_random_z(uint32_t &s0, uint32_t &s1) {
    return 3.0f - __uint_as_float(next_random_word(s0, s1));
}

// .cl:173-201 in two halves, g = {centre, radius^2}: discriminant first, roots second.
struct HitPre {
    float b, det;
};
RT_DEV HitPre hit_pre(float4 g, V3 o, V3 d) {
    V3 op = mk(g.x - o.x, g.y - o.y, g.z - o.z);
#if RT_OPT_EXACT_DECISIONS
    float b = dot_decision(op, d);
    return HitPre{ b, mul_decision(b, b) - dot_decision(op, op) + g.w };
#else
    float b = dot(op, d);
    return HitPre{ b, b * b - dot(op, op) + g.w };
#endif
}
RT_DEV float hit_post(HitPre p) {
    float sq = rt_sqrt_det(p.det);
    float t1 = p.b - sq;
    float t2 = p.b + sq;
    float t = t1 > RT_EPS ? t1 : (t2 > RT_EPS ? t2 : 0.f);
    return p.det < 0.f ? 0.f : t;
}
// The same decision with fewer operations, for the sweeps: `hit` says that the reference's
// SphereIntersect returns a non-zero distance and `t` is that distance.  sq >= 0 makes
// t1 = fl(b - sq) <= b <= fl(b + sq) = t2, so "t1 > EPS or t2 > EPS" is "t2 > EPS".  The sign test
// of the discriminant stays (it is the comparison the wave ballot of the sweep already made): the
// unscaled v_sqrt_f32 does not return NaN for a negative subnormal.
struct HitRoots {
    float t;
    bool hit;
};
RT_DEV HitRoots hit_roots(HitPre p) {
    const float sq = rt_sqrt_det(p.det);
    const float t1 = p.b - sq;
    const float t2 = p.b + sq;
    return HitRoots{ t1 > RT_EPS ? t1 : t2, (p.det >= 0.f) && (t2 > RT_EPS) };
}
#if RT_OPT_GLOBAL_TABLES && !RT_OPT_WALK
// Records of a table in HBM / L2 at a wave-uniform address -- the kernel's argument plus a loop counter -- fetched through the SCALAR cache into scalar
// registers: one s_load_dwordx16 per four records, no vector-memory instruction, no address per lane, no vector register, and the test takes the record's
// words as scalar operands.  Written as the instruction: left to itself the compiler issues the same load, but sinks it to where its result is first used,
// behind the tests it was meant to overlap (the use is the loop's back edge).  So the request and the wait are two statements: request_four_uniform leaves
// the registers pending, arrived() is the s_waitcnt before anything may read them -- on EVERY way out of a trip: a pending load writes its registers
// whoever owns them by then.  The tables are written by an earlier kernel on the stream (rt_build_tables_kernel); the scalar cache is invalidated at the
// start of every kernel.
typedef float F16 __attribute__((ext_vector_type(16)));
RT_DEV F16 request_four_uniform(const float4 *p) {
    F16 r;
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(r) : "s"(p));
    return r;
}
RT_DEV F16 arrived(F16 v) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v));
    return v;
}
RT_DEV float4 record_of(F16 v, int k) { return make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]); }
#endif
// true when some active lane needs the roots (NaN discriminants never hit: .cl:185-200)
RT_DEV bool wave_any_nonneg(float det) { return __builtin_amdgcn_ballot_w64(det >= 0.f) != 0ull; }
// closest hit over spheres [0, n): .cl:215-232
RT_DEV void sweep_closest(const float4 *s_geom, uint32_t n, V3 o, V3 d, float &t, uint32_t &id,
                          unsigned long long &roots) {
    uint32_t i = 0;
#if RT_OPT_GLOBAL_TABLES && !RT_OPT_WALK
    // The table lies in HBM / L2 (rt_trace_*_g: no hierarchy and more records than the sweep stages -- 40 KB, four workgroups per CU: rt_launch.hip).  Every lane tests the SAME record, so the table goes through
    // the scalar cache, the next four records requested before these four are tested.  (As per-lane loads of one address the sweep was bound by the
    // texture-address unit, 16 cycles per record and CU against the 8 its four SIMDs need for the test: vector ALU 15 % busy; that form, two records per
    // scalar load and a per-wavefront LDS window filled by all 64 lanes were measured against this one: profiles/r06_g_sweep_forms.jsonl.)  The same
    // tests in the same order.
    if (n >= 4) {
        F16 a = arrived(request_four_uniform(s_geom));
        for (;;) {
            const bool more = i + 8 <= n;
            const F16 b = request_four_uniform(s_geom + (more ? i + 4 : i));      // (the last group asks for itself again: no read beyond the table)
            const HitPre p0 = hit_pre(record_of(a, 0), o, d), p1 = hit_pre(record_of(a, 1), o, d), p2 = hit_pre(record_of(a, 2), o, d), p3 = hit_pre(record_of(a, 3), o, d);
            if (wave_any_nonneg(p0.det)) { roots += 1; const HitRoots h = hit_roots(p0); if (h.hit && h.t < t) { t = h.t; id = i; } }
            if (wave_any_nonneg(p1.det)) { roots += 1; const HitRoots h = hit_roots(p1); if (h.hit && h.t < t) { t = h.t; id = i + 1; } }
            if (wave_any_nonneg(p2.det)) { roots += 1; const HitRoots h = hit_roots(p2); if (h.hit && h.t < t) { t = h.t; id = i + 2; } }
            if (wave_any_nonneg(p3.det)) { roots += 1; const HitRoots h = hit_roots(p3); if (h.hit && h.t < t) { t = h.t; id = i + 3; } }
            i += 4;
            a = arrived(b);
            if (!more) break;
        }
    }
#endif
    for (; i + 2 <= n; i += 2) {
        const float4 g0 = s_geom[i], g1 = s_geom[i + 1];
        const HitPre p0 = hit_pre(g0, o, d), p1 = hit_pre(g1, o, d);
        if (wave_any_nonneg(p0.det)) {
            roots += 1;
            const HitRoots h0 = hit_roots(p0);
            if (h0.hit && h0.t < t) { t = h0.t; id = i; }
        }
        if (wave_any_nonneg(p1.det)) {
            roots += 1;
            const HitRoots h1 = hit_roots(p1);
            if (h1.hit && h1.t < t) { t = h1.t; id = i + 1; }
        }
    }
    for (; i < n; ++i) {
        const HitPre p0 = hit_pre(s_geom[i], o, d);
        if (wave_any_nonneg(p0.det)) {
            roots += 1;
            const HitRoots h0 = hit_roots(p0);
            if (h0.hit && h0.t < t) { t = h0.t; id = i; }
        }
    }
}

// any hit closer than max_t, .cl:234-247.  Returns the index of the first blocking sphere, or n.
// A lane stops looking at its first hit; the wavefront leaves when every active lane has one.
RT_DEV uint32_t sweep_any(const float4 *s_geom, uint32_t n, V3 o, V3 d, float max_t,
                          unsigned long long &roots) {
    uint32_t first = n;
    uint32_t i = 0;
#if RT_OPT_GLOBAL_TABLES && !RT_OPT_WALK
    if (n >= 4) {                   // (as in sweep_closest: four records per scalar load, the next four in flight meanwhile)
        F16 a = arrived(request_four_uniform(s_geom));
        for (;;) {
            const bool more = i + 8 <= n;
            const F16 b = request_four_uniform(s_geom + (more ? i + 4 : i));
            const HitPre p0 = hit_pre(record_of(a, 0), o, d), p1 = hit_pre(record_of(a, 1), o, d), p2 = hit_pre(record_of(a, 2), o, d), p3 = hit_pre(record_of(a, 3), o, d);
            if (__builtin_amdgcn_ballot_w64(first == n && p0.det >= 0.f) != 0ull) { roots += 1; const HitRoots h = hit_roots(p0); if (first == n && h.hit && h.t < max_t) first = i; }
            if (__builtin_amdgcn_ballot_w64(first == n && p1.det >= 0.f) != 0ull) { roots += 1; const HitRoots h = hit_roots(p1); if (first == n && h.hit && h.t < max_t) first = i + 1; }
            if (__builtin_amdgcn_ballot_w64(first == n && p2.det >= 0.f) != 0ull) { roots += 1; const HitRoots h = hit_roots(p2); if (first == n && h.hit && h.t < max_t) first = i + 2; }
            if (__builtin_amdgcn_ballot_w64(first == n && p3.det >= 0.f) != 0ull) { roots += 1; const HitRoots h = hit_roots(p3); if (first == n && h.hit && h.t < max_t) first = i + 3; }
            a = arrived(b);
            if (__builtin_amdgcn_ballot_w64(first == n) == 0ull) return first;
            i += 4;
            if (!more) break;
        }
    }
#endif
    for (; i + 2 <= n; i += 2) {
        const float4 g0 = s_geom[i], g1 = s_geom[i + 1];
        const HitPre p0 = hit_pre(g0, o, d), p1 = hit_pre(g1, o, d);
        // a lane that already has its blocker asks for no more roots
        if (__builtin_amdgcn_ballot_w64(first == n && p0.det >= 0.f) != 0ull) {
            roots += 1;
            const HitRoots h0 = hit_roots(p0);
            if (first == n && h0.hit && h0.t < max_t) first = i;
        }
        if (__builtin_amdgcn_ballot_w64(first == n && p1.det >= 0.f) != 0ull) {
            roots += 1;
            const HitRoots h1 = hit_roots(p1);
            if (first == n && h1.hit && h1.t < max_t) first = i + 1;
        }
        if (__builtin_amdgcn_ballot_w64(first == n) == 0ull) return first;
    }
    for (; i < n; ++i) {
        const HitPre p0 = hit_pre(s_geom[i], o, d);
        if (__builtin_amdgcn_ballot_w64(first == n && p0.det >= 0.f) != 0ull) {
            roots += 1;
            const HitRoots h0 = hit_roots(p0);
            if (first == n && h0.hit && h0.t < max_t) first = i;
            if (__builtin_amdgcn_ballot_w64(first == n) == 0ull) return first;
        }
    }
    return first;
}

// .cl:438 and .cl:470: the two quantities a glass hit DECIDES on (total internal reflection; reflect or refract)
RT_DEV float cos2t_of(float nnt, float ddn) {
#if RT_OPT_EXACT_DECISIONS
    return 1.f - mul_decision(mul_decision(nnt, nnt), 1.f - mul_decision(ddn, ddn));
#else
    return 1.f - nnt * nnt * (1.f - ddn * ddn);
#endif
}
RT_DEV float roulette_p(float Re) {
#if RT_OPT_EXACT_DECISIONS
    return .25f + mul_decision(.5f, Re);
#else
    return .25f + .5f * Re;
#endif
}

// One light of SampleLights (.cl:258-295) up to the visibility test.  Draws its two numbers
// whatever happens next, as the reference does.  Returns true when a shadow ray is needed and
// then gives its unit direction, its length and the numerator 4*pi*r^2*wi*wo of .cl:297.
RT_DEV bool sample_light(float4 la, float4 lb, uint32_t &s0, uint32_t &s1, uint32_t &c_draws, V3 hp, V3 nl,
                         V3 &sd, float &len, float &numer) {
    float zc = next_random_z(s0, s1);                                      // .cl:203-213: 1 - 2 u1
    float u2 = next_random(s0, s1);
    c_draws += 2;
    float ring = rt_sqrt_unit(fmaxf(0.f, 1.f - zc * zc));
    float sphi, cphi;
#if RT_FAST
    fm_sincos_turns(u2, sphi, cphi);
#else
    dm_sincosf_pos((2.f * RT_PI) * u2, sphi, cphi);
#endif
    V3 us = mk(ring * cphi, ring * sphi, zc);
    V3 on_light = add(scale(us, la.w), mk(la.x, la.y, la.z));
    sd = sub(on_light, hp);
    sd = scale(sd, sqrt_and_rcp(dot(sd, sd), len));
    float wo = dot_decision(sd, us);
    if (wo > 0.f) return false;                                            // far side of the light
    wo = -wo;
    float wi = dot_decision(sd, nl);
    numer = lb.w * wi * wo;
    return wi > 0.f;
}

#if RT_OPT_COOP
constexpr int kCoopSlots = 32;                          // pending rays a wavefront can share out
constexpr int kCoopWaveFloats = kCoopSlots * 9;         // per wavefront: 2 float4 + 1 word per slot

// Cooperative any-hit (.cl:234-247) for the `want` lanes of a wavefront.  Called by every lane
// that is in the loop.  K = pending rays (wave ballot); with G = 64 / K >= 2 each ray is tested
// by G lanes, lane g of a ray taking spheres g, g+G, g+2G, ...; the smallest blocking index per
// ray is combined with an LDS atomic min, so the result (and the test count derived from it)
// equals the sequential sweep.  G counts only the lanes that are present in this call (a wave
// ballot of the callers).  With G < 2 the lanes sweep for themselves.
RT_DEV uint32_t coop_any(const float4 *s_geom, uint32_t n, bool want, V3 o, V3 d, float max_t, float *scratch, int kmax) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(want);
    const int K = __popcll(m);
    if (K == 0) return n;
    // lanes that are here to help: finished or gated lanes of the wavefront are not
    const unsigned long long present = __builtin_amdgcn_ballot_w64(true);
    const int G = __popcll(present) / K;
    unsigned long long unused_roots = 0;
    if (G < 2 || (kmax > 0 && K > kmax)) {
        uint32_t first = n;
        if (want) first = sweep_any(s_geom, n, o, d, max_t, unused_roots);
        return first;
    }
    float4 *ray_a = reinterpret_cast<float4 *>(scratch);                   // {origin, max_t}
    float4 *ray_b = ray_a + kCoopSlots;                                    // {direction, -}
    uint32_t *res = reinterpret_cast<uint32_t *>(ray_b + kCoopSlots);      // min blocking index
    const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (want) {
        ray_a[rank] = make_float4(o.x, o.y, o.z, max_t);
        ray_b[rank] = make_float4(d.x, d.y, d.z, 0.f);
        res[rank] = n;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int idx = __builtin_amdgcn_mbcnt_hi((uint32_t)(present >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)present, 0u));
    const int slot = idx % K;
    const int grp = idx / K;
    if (grp < G) {
        const float4 ra = ray_a[slot];
        const float4 rb = ray_b[slot];
        const V3 ro = mk(ra.x, ra.y, ra.z), rd = mk(rb.x, rb.y, rb.z);
        uint32_t found = n;
        for (uint32_t i = (uint32_t)grp; i < n; i += (uint32_t)G) {
            const HitPre p0 = hit_pre(s_geom[i], ro, rd);
            if (wave_any_nonneg(p0.det)) {
                const HitRoots h0 = hit_roots(p0);
                if (h0.hit && h0.t < ra.w) {
                    found = i;
                    break;
                }
            }
        }
        if (found < n) atomicMin(&res[slot], found);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t first = n;
    if (want) first = res[rank];
    __builtin_amdgcn_wave_barrier();
    return first;
}
#endif


// .cl:34
RT_DEV int to_int(float v) {
    float c = fminf(fmaxf(v, 0.f), 1.f);
#if RT_FAST
    float g = fm_powf(c, 1.f / 2.2f);
#else
    float g = dm_powf(c, 1.f / 2.2f);
#endif
    return (int)(g * 255.f + .5f);
}

RT_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

#if RT_OPT_WALK
#include "rt_walk.inc.h"       // large scenes: the walk of the hierarchy as lane state (its own kernel body)
#elif !defined(RT_NO_RENDER_KERNEL)     // (the first instantiation of a product library carries the helpers and the pack kernel only)
extern "C" __global__ void __launch_bounds__(64 * RT_OPT_WG_WAVES, RT_OPT_MINWAVES) RT_KERNEL_NAME(const LaunchParams P) {
    constexpr int kBlockThreads = 64 * RT_OPT_WG_WAVES;      // (shadow the 4-wavefront constants of rt_device.h)
    constexpr int kTileW = 8 * RT_OPT_WG_WAVES;
    (void)kTileW;
    extern __shared__ float4 lds[];
    const uint32_t n = P.scene.n_spheres;
    const uint32_t n_lights = P.scene.n_lights;
#if RT_OPT_GLOBAL_TABLES
    const float4 *s_geom = P.scene.geom;     // nothing staged but the reciprocals of the running average
#else
    float4 *s_geom = lds;
    float4 *s_lightA = s_geom + n;           // {centre, radius}
    float4 *s_lightB = s_lightA + n_lights;       // {emission, 4*pi*radius^2}
    float4 *s_emis = s_lightB + n_lights;         // {emission, bits(refl)}   (if mat_in_lds)
    float4 *s_colr = s_emis + n;             // {colour, radius}
#endif
    // 1/(s+1) of the running average (.cl:585), one IEEE division per sample index per
    // workgroup instead of one per lane per sample
#if RT_OPT_GLOBAL_TABLES
    float *s_k2 = reinterpret_cast<float *>(lds);       // (rt_launch.hip bind_tables: an instance over tables in HBM / L2 never has mat_in_lds)
#else
    float *s_k2 = reinterpret_cast<float *>(P.mat_in_lds ? s_colr + n : s_emis);
#endif
    const bool k2_in_lds = P.n_samples <= kMaxK2Table;

    const int tid = threadIdx.x;
    __shared__ unsigned long long s_stat[5];
    __shared__ unsigned s_tile_cost;
    __shared__ unsigned long long s_wg_t0;      // the workgroup's start on the device's wall clock (10 ns ticks)
    (void)s_tile_cost;
    (void)s_wg_t0;
    if (tid < 5) s_stat[tid] = 0;
    if (tid == 5) {
        s_tile_cost = 0u;
        s_wg_t0 = __builtin_amdgcn_s_memrealtime();
    }
#if !RT_OPT_GLOBAL_TABLES
    for (uint32_t i = tid; i < n; i += kBlockThreads) s_geom[i] = P.scene.geom[i];
    for (uint32_t i = tid; i < n_lights; i += kBlockThreads) {
        s_lightA[i] = P.scene.lightA[i];
        s_lightB[i] = P.scene.lightB[i];
    }
#endif
#if !RT_OPT_GLOBAL_TABLES
    if (P.mat_in_lds) {
        for (uint32_t i = tid; i < n; i += kBlockThreads) {
            s_emis[i] = P.scene.emis[i];
            s_colr[i] = P.scene.colr[i];
        }
    }
#endif
    if (k2_in_lds)
        for (int i = tid; i < P.n_samples; i += kBlockThreads)
            s_k2[i] = rt_rcp((float)(P.first_sample + i) + 1.f);
    __syncthreads();

    // ---- pixel of this lane ---------------------------------------------------------
    const int wave = tid >> 6, lane = tid & 63;
    (void)wave;
#if RT_OPT_TIMELOG
    const unsigned long long tl_start = __builtin_amdgcn_s_memrealtime();
    if (P.timelog && tid == 0) atomicMin(&P.timelog[8 * (size_t)P.seq], tl_start);
    if (P.timelog && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
        P.timelog[8 * (size_t)P.seq + 2] = 1ull;
        P.timelog[8 * (size_t)P.seq + 3] = P.tl_tag;
    }
#endif
#if RT_OPT_PERSIST
    // pixels are handed out inside the loop; nothing is owned yet
    int x = 0, lrow = 0, y = 0;
    bool has_pixel = false;
    size_t gid = 0, ci = 0;
    uint32_t s0 = 0, s1 = 0;
    V3 acc = mk(0.f, 0.f, 0.f);
    const int s_end = P.first_sample + P.n_samples;
    int s = s_end;
    uint32_t cur_tile = 0, cur_used = 64;      // wave-uniform: current 8x8 tile, pixels handed out of it
    uint32_t c_samples = 0;
#else
    // which 32x8 tile this workgroup renders: its own index, or the next one of the heavy-first order
    const unsigned block_linear = blockIdx.x + blockIdx.y * gridDim.x;
    const unsigned tile_id = P.order ? P.order[block_linear] : block_linear;           // (wave-uniform: a scalar load)
    const int tile_by = (int)(tile_id / gridDim.x), tile_bx = (int)(tile_id - (unsigned)tile_by * gridDim.x);
    // The pixel of this lane: the 8x8 square of its wavefront.  (Which lane renders which pixel is invisible in the results: a pixel's
    // samples, draws and arithmetic are its own.  Rounds 2-4 also DEALT the pixels of 32x32 regions to wavefronts by the cost the last
    // launch had left for them; on passes the costs had not seen that gained 0 .. 2 % when the costs were fresh and lost 7 % on a moving
    // scene, where they never are: removed in round 4, profiles/r04q_pixel_deal_*.)
    const int x = tile_bx * kTileW + wave * 8 + (lane & 7), lrow = tile_by * kTileH + (lane >> 3);
    const int tile = lrow / P.tile_rows;
    const int y = (tile * P.nranks + P.rank) * P.tile_rows + (lrow - tile * P.tile_rows);
    const bool valid = (x < P.w) && (lrow < P.local_rows) && (y < P.h);
    // What stays in registers through the loop of the pixel's place: x | y << 16 in ONE register (the camera ray needs
    // both per sample; the host refuses images beyond 65535 in either direction).  The local row, the validity and the
    // 64-bit indices are formed again after the loop from the lane number and the tile.
    const uint32_t xy = (uint32_t)x | ((uint32_t)y << 16);

    uint32_t s0 = 0, s1 = 0;
    V3 acc = mk(0.f, 0.f, 0.f);
    int s = P.first_sample;
    const int s_end = valid ? P.first_sample + P.n_samples : P.first_sample;
    if (valid) {
        const size_t gid = (size_t)y * (size_t)P.w + (size_t)x;             // .cl:560-563
        const size_t ci = (size_t)(P.h - y - 1) * (size_t)P.w + (size_t)x;  // .cl:579
        const uint2 sd = *reinterpret_cast<const uint2 *>(P.seeds_in + 2 * gid);
        s0 = sd.x;
        s1 = sd.y;
        if (P.first_sample > 0) acc = mk(P.colors[3 * ci], P.colors[3 * ci + 1], P.colors[3 * ci + 2]);
    }
#endif


    uint32_t c_closest = 0, c_shadow = 0, c_draws = 0;
    unsigned long long c_tests = 0;   // shadow-ray tests (per-ray additions, never inside a sphere loop); closest-hit rays add n each, at the end

    // ---- path state -----------------------------------------------------------------
    V3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 1.f);
    V3 thr = mk(1.f, 1.f, 1.f), rad = mk(0.f, 0.f, 0.f);
    int depth = 0;
    bool after_specular = true;
    bool need_ray = true;

    unsigned long long st_roots_c = 0, st_roots_s = 0;   // wave-uniform; dead unless RT_OPT_STAMPS
    (void)st_roots_s;
#if RT_OPT_COOP
    __shared__ __attribute__((aligned(16))) float s_coop[RT_OPT_WG_WAVES * kCoopWaveFloats];
#endif
#if RT_OPT_STAMPS
    __shared__ unsigned long long s_census[12];
    if (tid < 12) s_census[tid] = 0;
    __syncthreads();
#endif
    for (;;) {
        RT_STAMP(8);
        // The wave-level decisions below (ballots) are only meaningful if ALL lanes that are
        // still in the loop execute them together.  A `continue` for the waiting lanes gives the
        // loop a second back edge, and the compiler is then free to spin the waiting lanes
        // through the header on their own (observed: lanes at different trip counts, ballots
        // that see a subset).  So waiting lanes skip the body through one `if` instead.
        bool idle = false;
        // Gated regeneration: lanes whose path has ended wait until P.regen_gate of them can
        // start together (or nothing else is in flight).  Free-running lanes (gate 1) drift
        // apart in phase, so that every section of the loop runs in almost every trip for a
        // fraction of the lanes; a small gate keeps the lanes of a coherent tile in the same
        // phase of the bounce loop at the price of a few idle lane-trips.  Ordering within a
        // pixel is untouched.
#if RT_OPT_PERSIST
        {
            // lanes between paths; those without samples left also want a new pixel
            const unsigned long long bw = __builtin_amdgcn_ballot_w64(need_ray);
            const unsigned long long ba = __builtin_amdgcn_ballot_w64(!need_ray);
            const bool go = (__popcll(bw) >= P.regen_gate) || (ba == 0ull);
            const bool take = need_ray && go && s >= s_end;
            const unsigned long long F = __builtin_amdgcn_ballot_w64(take);
            if (F != 0ull) {
                if (take && has_pixel) {                                   // .cl:580-599 of the finished pixel
                    P.colors[3 * ci] = acc.x;
                    P.colors[3 * ci + 1] = acc.y;
                    P.colors[3 * ci + 2] = acc.z;
                    if (!(P.skip_pixels & 1))
                        P.pixels[(size_t)lrow * (size_t)P.w + (size_t)x] =
                            (uint32_t)(to_int(acc.x) | (to_int(acc.y) << 8) | (to_int(acc.z) << 16));
                    P.seeds[2 * gid] = s0;
                    P.seeds[2 * gid + 1] = s1;
                    c_samples += (uint32_t)P.n_samples;
                    has_pixel = false;
                }
                // hand out the next popcount(F) pixels of the wavefront's tile stream
                const uint32_t n_take = (uint32_t)__popcll(F);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(F >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)F, 0u));
                uint32_t next_tile = cur_tile;
                if (cur_used + n_take > 64u) {                             // wave-uniform: pull a tile
                    const unsigned long long here = __builtin_amdgcn_ballot_w64(true);
                    const int leader = __ffsll((long long)here) - 1;
                    uint32_t got = 0;
                    if (lane == leader) got = (uint32_t)atomicAdd(&P.counters[30], 1ull);
                    next_tile = (uint32_t)__shfl((int)got, leader, 64);
                }
                const uint32_t slot = cur_used + rank;
                const uint32_t my_tile = slot < 64u ? cur_tile : next_tile;
                const uint32_t my_idx = slot & 63u;
                if (cur_used + n_take > 64u) {
                    cur_used = cur_used + n_take - 64u;
                    cur_tile = next_tile;
                } else {
                    cur_used += n_take;
                }
                if (take) {
                    if (my_tile >= (uint32_t)P.n_tiles) break;            // queue drained: this lane retires
                    const int tx = (int)(my_tile % (uint32_t)P.tiles_x), ty = (int)(my_tile / (uint32_t)P.tiles_x);
                    x = tx * 8 + (int)(my_idx & 7u);
                    lrow = ty * 8 + (int)(my_idx >> 3);
                    const int rtile = lrow / P.tile_rows;
                    y = (rtile * P.nranks + P.rank) * P.tile_rows + (lrow - rtile * P.tile_rows);
                    if ((x < P.w) && (lrow < P.local_rows) && (y < P.h)) {
                        gid = (size_t)y * (size_t)P.w + (size_t)x;         // .cl:560-563
                        ci = (size_t)(P.h - y - 1) * (size_t)P.w + (size_t)x;   // .cl:579
                        s0 = P.seeds_in[2 * gid];
                        s1 = P.seeds_in[2 * gid + 1];
                        acc = mk(0.f, 0.f, 0.f);
                        if (P.first_sample > 0) acc = mk(P.colors[3 * ci], P.colors[3 * ci + 1], P.colors[3 * ci + 2]);
                        s = P.first_sample;
                        has_pixel = true;
                    }
                }
            }
            idle = need_ray && (!go || s >= s_end);                        // gated, or drew a pixel outside the image
        }
        if (!idle) {   // (no `continue`: every lane must meet again at the loop top, see below)
        if (need_ray) {
#else
        if (need_ray && s >= s_end) break;
        if (P.regen_gate > 1) {
            const unsigned long long bw = __builtin_amdgcn_ballot_w64(need_ray);
            const unsigned long long ba = __builtin_amdgcn_ballot_w64(!need_ray);
            const bool go = (__popcll(bw) >= P.regen_gate) || (ba == 0ull);
            idle = need_ray && !go;
        }
        if (!idle) {   // (no `continue`: every lane must meet again at the loop top, see below)
        if (need_ray) {
#endif
            RT_STAMP(0);
            // ---- camera ray, .cl:494-549 ----
            // The camera (12 floats) and 1/w, 1/h are read from the kernel-argument segment HERE, once per sample, through
            // the scalar cache, instead of occupying 14 scalar registers through the whole loop: the loop overfills the
            // scalar file, its spills go to lanes of a vector register, and that register was the one the allocator then
            // lacked (a private segment of 16 bytes in the cooperative instances).
            const volatile __attribute__((address_space(4))) LaunchParams *cp =
                (const volatile __attribute__((address_space(4))) LaunchParams *)__builtin_amdgcn_kernarg_segment_ptr();
            const float inv_w = cp->inv_w, inv_h = cp->inv_h;                // .cl:503-504, divided on the host
            const V3 cam_o = mk(cp->cam.orig.x, cp->cam.orig.y, cp->cam.orig.z);
            const V3 cam_d = mk(cp->cam.dir.x, cp->cam.dir.y, cp->cam.dir.z);
            const V3 cam_x = mk(cp->cam.x.x, cp->cam.x.y, cp->cam.x.z);
            const V3 cam_y = mk(cp->cam.y.x, cp->cam.y.y, cp->cam.y.z);
            float j1 = next_random_centred(s0, s1);
            float j2 = next_random_centred(s0, s1);
            c_draws += 2;
#if RT_OPT_PERSIST
            float kcx = ((float)x + j1) * inv_w - 0.5f;
            float kcy = ((float)y + j2) * inv_h - 0.5f;
#else
            // (unpacked HERE, per sample: the compiler otherwise hoists x and y out of the loop into two more registers --
            // the ones the 4-wavefront cooperative instance then spilled)
            uint32_t xy_now = xy;
            asm volatile("; pixel coordinates unpacked per sample" : "+v"(xy_now));
            float kcx = ((float)(xy_now & 0xffffu) + j1) * inv_w - 0.5f;
            float kcy = ((float)(xy_now >> 16) + j2) * inv_h - 0.5f;
#endif
            V3 rd = mk(cam_x.x * kcx + cam_y.x * kcy + cam_d.x,
                       cam_x.y * kcx + cam_y.y * kcy + cam_d.y,
                       cam_x.z * kcx + cam_y.z * kcy + cam_d.z);
            o = add(scale(rd, 0.1f), cam_o);
            d = unit(rd);
            thr = mk(1.f, 1.f, 1.f);
            rad = mk(0.f, 0.f, 0.f);
            depth = 0;
            after_specular = true;
            need_ray = false;
        }

        // ---- closest hit, .cl:215-232: wave-uniform sweep, LDS broadcast reads ----
        float t = 1e20f;
        uint32_t id = 0;
        RT_STAMP(1);
        st_roots_c = 0;
        sweep_closest(s_geom, n, o, d, t, id, st_roots_c);
        RT_STAMP_ROOTS(10, st_roots_c);
        c_closest += 1;

        bool path_done = false;
        bool is_diff = false, is_gloss = false;   // what the hit asks for next
        // the hit record: written by the hit branch, read only where is_diff / is_gloss say it was
        V3 hp, nrm, nl, col;
        float dp;
        int refl;
        if (!(t < 1e20f)) {
            path_done = true;                                              // miss, .cl:327-330
        } else {
            RT_STAMP(2);
            const float4 ge = s_geom[id];
            float4 em4, co4;
#if RT_OPT_GLOBAL_TABLES
            {
                // (rt_trace_*_g: the two pointers are read from the kernel-argument segment HERE, once per hit, as the camera is once per sample:
                // the sweep keeps two pairs of records in scalar registers, and these four would be spilled to hold them)
                const volatile __attribute__((address_space(4))) LaunchParams *mp =
                    (const volatile __attribute__((address_space(4))) LaunchParams *)__builtin_amdgcn_kernarg_segment_ptr();
                em4 = mp->scene.emis[id];
                co4 = mp->scene.colr[id];
            }
#else
            if (P.mat_in_lds) {                 // wave-uniform: ds_read / global_load, not flat_load
                em4 = s_emis[id];
                co4 = s_colr[id];
                asm volatile("; materials from LDS" : "+v"(em4.x));   // keeps the two loads from being merged into one generic-pointer load
            } else {
                em4 = P.scene.emis[id];
                co4 = P.scene.colr[id];
            }
#endif
            const V3 em = mk(em4.x, em4.y, em4.z);
            col = mk(co4.x, co4.y, co4.z);
            refl = __float_as_int(em4.w);

            hp = add(o, scale(d, t));                                      // .cl:338-340
            nrm = unit(sub(hp, mk(ge.x, ge.y, ge.z)));                     // .cl:345-347
            dp = dot(nrm, d);
            nl = scale(nrm, -1.f * cl_sign(dp));                           // .cl:354-355

            if (!((em.x == 0.f) && (em.z == 0.f))) {                       // .cl:358-368
                if (after_specular) rad = add(rad, mul(thr, scale(em, fabsf(dp))));
                path_done = true;
            } else if (refl == RT_DIFF) {                                  // .cl:370-373
                after_specular = false;
                thr = mul(thr, col);
                is_diff = true;
            } else {
                is_gloss = true;
            }
        }

        // ---- next-event estimation, .cl:249-303 ----
        V3 ld = mk(0.f, 0.f, 0.f);
#if RT_OPT_COOP
        // The light loop runs at wavefront level so that lanes without a shadow ray of their
        // own (not diffuse, far side of the light, facing away) can take part of the any-hit
        // sweep of the lanes that have one: coop_any splits the sphere list of each pending
        // ray over 64/K lanes (K = rays pending in the wavefront, from a wave ballot).
        if (__builtin_amdgcn_ballot_w64(is_diff) != 0ull) {
            for (uint32_t j = 0; j < n_lights; ++j) {
                bool want = false;
                V3 sd = mk(0.f, 0.f, 1.f);
                float len = 1.f, numer = 0.f;
                const float4 lb = s_lightB[j];
                if (is_diff) {
                    RT_STAMP(3);
                    want = sample_light(s_lightA[j], lb, s0, s1, c_draws, hp, nl, sd, len, numer);
                }
                RT_STAMP(4);
                const uint32_t first = coop_any(s_geom, n, want, hp, sd, len - RT_EPS, s_coop + wave * kCoopWaveFloats, P.coop_kmax);
#if RT_OPT_COOP == 2
                {   // verification instance: the sequential sweep beside the cooperative one
                    unsigned long long dummy = 0;
                    const unsigned long long mm = __builtin_amdgcn_ballot_w64(want);
                    if (want) {
                        const uint32_t ref_first = sweep_any(s_geom, n, hp, sd, len - RT_EPS, dummy);
                        atomicAdd(&P.counters[20], 1ull);
                        if (ref_first != first) {
                            atomicAdd(&P.counters[21], 1ull);
                            P.counters[22] = ((unsigned long long)__popcll(mm) << 48) | ((unsigned long long)first << 24) | ref_first;
                            {
                                const int K_ = __popcll(mm), G_ = 64 / K_;
                                const int rank_ = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                                const float4 g_ = s_geom[ref_first];
                                const HitPre pp = hit_pre(g_, hp, sd);
                                P.counters[23] = ((unsigned long long)rank_ << 48) | ((unsigned long long)(ref_first % G_) << 32) |
                                                 (unsigned long long)__float_as_uint(pp.det);
                            }
                        }
                    }
                }
#endif
                if (want) {
                    c_shadow += 1;
                    const bool blocked = first < n;
                    c_tests += blocked ? first + 1 : n;
                    if (!blocked) {
                        float k = rt_div(numer, len * len);                // .cl:297
                        ld = add(ld, scale(mk(lb.x, lb.y, lb.z), k));
                    }
                }
            }
        }
#else
        if (is_diff) {
            for (uint32_t j = 0; j < n_lights; ++j) {
                RT_STAMP(3);
#if RT_OPT_GLOBAL_TABLES
                const volatile __attribute__((address_space(4))) LaunchParams *lp =           // (as the material pointers above: once per light)
                    (const volatile __attribute__((address_space(4))) LaunchParams *)__builtin_amdgcn_kernarg_segment_ptr();
                const float4 lb = lp->scene.lightB[j];
                const float4 la = lp->scene.lightA[j];
#else
                const float4 lb = s_lightB[j];
                const float4 la = s_lightA[j];
#endif
                V3 sd;
                float len, numer;
                if (!sample_light(la, lb, s0, s1, c_draws, hp, nl, sd, len, numer)) continue;
                // ---- shadow ray, any hit, .cl:234-247 ----
                c_shadow += 1;
                RT_STAMP(4);
                st_roots_s = 0;
                const uint32_t first = sweep_any(s_geom, n, hp, sd, len - RT_EPS, st_roots_s);
                RT_STAMP_ROOTS(11, st_roots_s);
                const bool blocked = first < n;
                c_tests += blocked ? first + 1 : n;
                if (!blocked) {
                    RT_STAMP(5);
                    float k = rt_div(numer, len * len);                    // .cl:297
                    ld = add(ld, scale(mk(lb.x, lb.y, lb.z), k));
                }
            }
        }
#endif

        if (is_diff) {
            rad = add(rad, mul(thr, ld));                                  // .cl:377-378
            RT_STAMP(6);
            // ---- cosine-weighted bounce, .cl:383-411 ----
            float u = next_random(s0, s1);
            float r2 = next_random(s0, s1);
            c_draws += 2;
            float r2s = rt_sqrt_unit(r2);
            V3 w = nl;
            V3 a = (fabsf(w.x) > .1f) ? mk(0.f, 1.f, 0.f) : mk(1.f, 0.f, 0.f);
            V3 uu = unit(cross(a, w));
            V3 vv = cross(w, uu);
            float s1v, c1v;
#if RT_FAST
            fm_sincos_turns(u, s1v, c1v);
#else
            dm_sincosf_pos((2.f * RT_PI) * u, s1v, c1v);
#endif
            V3 nd = add(scale(uu, c1v * r2s), scale(vv, s1v * r2s));
            nd = add(nd, scale(w, rt_sqrt_unit(1 - r2)));
            o = hp;
            d = nd;
        } else if (is_gloss) {
            // reflection direction shared by SPEC and REFR, .cl:416-419 / 428-431
            RT_STAMP(7);
            // n.d is the dp of the hit record; nl = n * m with m = -sign(dp), and for m = +-1 every
            // product and sum of (n.nl) and (d.nl) is the sign-flipped twin of the one in (n.n) and
            // (n.d) (rounding is sign-symmetric), so n.nl > 0 <=> dp < 0 and d.nl = m * dp = -|dp|.
            // dp = +-0 gives zeros of either sign, which the uses below do not tell apart; NaN stays NaN.
            const float n_dot_d = dp;
            V3 rfl = sub(d, scale(nrm, 2.f * n_dot_d));
            after_specular = true;
            if (refl == RT_SPEC) {                                         // .cl:413-424
                thr = mul(thr, col);
                d = rfl;
            } else {                                                       // .cl:425-489
                const bool into = dp < 0.f;
                const float ddn = -fabsf(dp);
                const float nc = 1.f, nt = 1.52f;
                float nnt = into ? nc / nt : nt / nc;
                float cos2t = cos2t_of(nnt, ddn);
                if (cos2t < 0.f) {                                         // total internal reflection
                    thr = mul(thr, col);
                    d = rfl;
                } else {
                    float kk = (into ? 1.f : -1.f) * (ddn * nnt + rt_sqrt(cos2t));
                    V3 td = unit(sub(scale(d, nnt), scale(nrm, kk)));
                    const float fa = nt - nc, fb = nt + nc;
                    const float R0 = fa * fa / (fb * fb);
                    float c = 1 - (into ? -ddn : dot(td, nrm));
                    float Re = R0 + (1 - R0) * c * c * c * c * c;
                    float Tr = 1.f - Re;
                    float Pr = roulette_p(Re);
                    float pick = next_random(s0, s1);
                    c_draws += 1;
                    // RP = Re / P and TP = Tr / (1 - P): only the branch taken is divided
                    const bool take_rfl = pick < Pr;
                    const float wgt = rt_div(take_rfl ? Re : Tr, take_rfl ? Pr : 1.f - Pr);
                    thr = mul(scale(thr, wgt), col);
                    d = take_rfl ? rfl : td;
                }
            }
            o = hp;
        }
        if (is_diff || is_gloss) {
            depth += 1;
            if (depth >= kMaxDepth) path_done = true;                      // .cl:320
        }

        if (path_done) {
            RT_STAMP(9);
            // ---- running average, .cl:580-589 ----
            if (s == 0) {
                acc = rad;
            } else {
                float k1 = (float)s;
                float k2 = k2_in_lds ? s_k2[s - P.first_sample] : rt_rcp((float)s + 1.f);
                acc = mk((acc.x * k1 + rad.x) * k2, (acc.y * k1 + rad.y) * k2,
                         (acc.z * k1 + rad.z) * k2);
            }
            s += 1;
            need_ray = true;
        }
        }   // if (!idle)
    }

#if !RT_OPT_PERSIST
    // The arguments the epilogue needs are read from the kernel-argument segment AGAIN here (a fresh
    // scalar load behind an opaque pointer) instead of staying live in SGPRs through the loop: the loop
    // already fills the scalar file, and keeping them cost SGPR spills and with them a private segment.
    const __attribute__((address_space(4))) LaunchParams *qp =
        (const __attribute__((address_space(4))) LaunchParams *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; epilogue arguments re-read" : "+s"(qp));
    const __attribute__((address_space(4))) LaunchParams &Q = *qp;
    // the lane's number from the execution mask (no register held for it): v_mbcnt of all ones
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool valid_e = s_end != Q.first_sample;       // (s_end was first_sample + n_samples for the lanes that own a pixel)
    if (valid_e && Q.n_samples > 0) {
        const int le = tile_by * kTileH + (lane_e >> 3);
        uint32_t xy_e = xy;
        asm volatile("; pixel coordinates unpacked after the loop" : "+v"(xy_e));     // (not before it, into registers held through it)
        const int xe = (int)(xy_e & 0xffffu), ye = (int)(xy_e >> 16);
        const size_t gid = (size_t)ye * (size_t)Q.w + (size_t)xe;           // .cl:560-563
        const size_t ci = (size_t)(Q.h - ye - 1) * (size_t)Q.w + (size_t)xe;   // .cl:579
        float *colors = Q.colors;
        colors[3 * ci] = acc.x;
        colors[3 * ci + 1] = acc.y;
        colors[3 * ci + 2] = acc.z;
        if (!(Q.skip_pixels & 1))                                          // (wave-uniform)
            Q.pixels[(size_t)le * (size_t)Q.w + (size_t)xe] =               // .cl:594-596
                (uint32_t)(to_int(acc.x) | (to_int(acc.y) << 8) | (to_int(acc.z) << 16));
        *reinterpret_cast<uint2 *>(Q.seeds + 2 * gid) = make_uint2(s0, s1);   // .cl:598-599
    }

#endif
    // ---- exact work counters: one atomic per counter per wavefront ----
#if RT_OPT_PERSIST
    uint32_t n_done = c_samples;
#else
    uint32_t n_done = valid_e ? (uint32_t)Q.n_samples : 0u;
#endif
    uint32_t t_samples = wave_sum(n_done);
    uint32_t t_closest = wave_sum(c_closest);
    uint32_t t_shadow = wave_sum(c_shadow);
    uint32_t t_draws = wave_sum(c_draws);
    unsigned long long tests64 = c_tests + (unsigned long long)c_closest * n;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tests64 += __shfl_xor(tests64, off, 64);
    // one LDS add per wavefront, then one global add per workgroup into one of kStatReplicas
    // separate lines (162 000 same-address global atomics cost 1.8 ms per launch: one word takes
    // about 88 atomics per microsecond)
#if !RT_OPT_PERSIST
    // what this tile cost: the wall clock of its slowest wavefront since the workgroup started (one start for the workgroup: a
    // start per wavefront needed the wavefront's number here, a register held through the loop for nothing else)
    if (lane_e == 0) atomicMax(&s_tile_cost, (unsigned)(__builtin_amdgcn_s_memrealtime() - s_wg_t0));   // 10 ns ticks
    const int lane_c = lane_e;
#else
    const int lane_c = lane;
#endif
    if (lane_c == 0) {
        atomicAdd(&s_stat[0], (unsigned long long)t_samples);
        atomicAdd(&s_stat[1], (unsigned long long)t_closest);
        atomicAdd(&s_stat[2], (unsigned long long)t_shadow);
        atomicAdd(&s_stat[3], tests64);
        atomicAdd(&s_stat[4], (unsigned long long)t_draws);
    }
    __syncthreads();
#if !RT_OPT_PERSIST
    // (bit 1 of the launch flags: a SHORT launch inside an accumulation window -- its cost is added to what the window's launches before it left,
    // so that a host which only ever launches a pass or two at a time gets its tiles ordered from 16 passes' worth of costs: rt_launch.hip)
    if (tid == 5 && Q.tile_cost) Q.tile_cost[tile_id] = (Q.skip_pixels & 2) ? Q.tile_cost[tile_id] + s_tile_cost : s_tile_cost;
#endif
    if (tid < 5) {
#if RT_OPT_PERSIST
        const unsigned block_linear = blockIdx.x + blockIdx.y * gridDim.x;
#endif
#if RT_OPT_PERSIST
        atomicAdd(&P.stats[(block_linear % (unsigned)kStatReplicas) * 8u + (unsigned)tid], s_stat[tid]);
#else
        atomicAdd(&Q.stats[(block_linear % (unsigned)kStatReplicas) * 8u + (unsigned)tid], s_stat[tid]);
#endif
    }
#if RT_OPT_STAMPS
    __syncthreads();
    if (tid < 12) atomicAdd(&P.counters[8 + tid], s_census[tid]);
#endif
#if RT_OPT_TIMELOG
    {
        const unsigned long long tl_end = __builtin_amdgcn_s_memrealtime();
        if (P.timelog && tid == 0) atomicMax(&P.timelog[8 * (size_t)P.seq + 1], tl_end);
        if (P.wavelog && lane == 0) {
            const size_t wi = ((size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 4 + (size_t)wave) * 3;
            uint32_t hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            P.wavelog[wi] = tl_start;
            P.wavelog[wi + 1] = tl_end;
            P.wavelog[wi + 2] = ((unsigned long long)xcc << 32) | hwid;
        }
    }
#endif
}
#endif   // !RT_OPT_WALK

#if !defined(RT_VARIANT_KERNEL)
// The packed frame from the colour plane (.cl:34,594-596) with this mode's toInt, for frames whose
// launches ran with the pixel store switched off (rt_set_pixel_write(ctx, 0) ... rt_read_pixels).
// One thread per local pixel; rows map to image rows as in the render kernel.
extern "C" __global__ void RT_PACK_KERNEL_NAME(const LaunchParams P) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int lrow = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int tile = lrow / P.tile_rows;
    const int y = (tile * P.nranks + P.rank) * P.tile_rows + (lrow - tile * P.tile_rows);
    if (x >= P.w || lrow >= P.local_rows || y >= P.h) return;
    const size_t ci = (size_t)(P.h - y - 1) * (size_t)P.w + (size_t)x;
    const float r = P.colors[3 * ci], g = P.colors[3 * ci + 1], b = P.colors[3 * ci + 2];
    P.pixels[(size_t)lrow * (size_t)P.w + (size_t)x] = (uint32_t)(to_int(r) | (to_int(g) << 8) | (to_int(b) << 16));
}
#endif

#if !RT_FAST && !defined(RT_VARIANT_KERNEL) && RT_DIAGNOSTICS
// every binary32 bit pattern: ieee_sqrt_lean against the compiler's correctly rounded sqrtf
extern "C" __global__ void rt_sqrt_check_kernel(unsigned long long *mismatches) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const uint32_t a = __float_as_uint(ieee_sqrt_lean(x)), r = __float_as_uint(sqrtf(x));
        const bool both_nan = ((a & 0x7fffffffu) > 0x7f800000u) && ((r & 0x7fffffffu) > 0x7f800000u);
        if (a != r && !both_nan) bad += 1;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// hit_post with the unchecked square root against hit_post with sqrtf: every discriminant in
// (0, 2^-96) and a few above, against b values around the decisions of the test
extern "C" __global__ void rt_hitpost_check_kernel(unsigned long long *mismatches) {
    const float bs[24] = { 0.f, -0.f, 0x1p-149f, 0x1p-126f, 0x1p-60f, 0x1p-48f, 0x1p-47f, 0x1p-46f, 0x1p-30f,
                           0x1p-24f, 0x1.fffffep-24f, 0x1p-23f, 0x1.000002p-23f, 0x1p-22f, 0.005f, 0x1.47ae12p-7f,
                           0.01f, 0x1.47ae16p-7f, 0.02f, 1.f, 3.f, 1000.f, 1e20f, 3e38f };
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < 0x10000000ull; k += stride) {
        const float det = __uint_as_float((uint32_t)k + 1u);          // (0, 2^-95)
        for (int j = 0; j < 24; ++j) {
            for (int sg = 0; sg < 2; ++sg) {
                const float b = sg ? -bs[j] : bs[j];
                const float sq_ref = sqrtf(det);
                const float t1r = b - sq_ref, t2r = b + sq_ref;
                const float tr = t1r > RT_EPS ? t1r : (t2r > RT_EPS ? t2r : 0.f);
                const float th = hit_post(HitPre{ b, det });
                const HitRoots hr = hit_roots(HitPre{ b, det });
                if (__float_as_uint(tr) != __float_as_uint(th)) bad += 1;
                if (__float_as_uint(tr) != __float_as_uint(hr.hit ? hr.t : 0.f)) bad += 1;
            }
        }
        // and the short decision (hit_roots) against the reference's, arbitrary bit patterns
        uint32_t h1 = (uint32_t)k * 2654435761u + 0x9e3779b9u, h2 = ((uint32_t)k ^ 0x85ebca6bu) * 2246822519u;
        h1 ^= h1 >> 15; h1 *= 2246822519u; h1 ^= h1 >> 13;
        h2 ^= h2 >> 16; h2 *= 3266489917u; h2 ^= h2 >> 14;
        const float rb = __uint_as_float(h1), rdet = __uint_as_float(h2);
        const float sq = sqrtf(rdet);
        const float t1 = rb - sq, t2 = rb + sq;
        float tr = t1 > RT_EPS ? t1 : (t2 > RT_EPS ? t2 : 0.f);
        tr = rdet < 0.f ? 0.f : tr;
        const HitRoots hr = hit_roots(HitPre{ rb, rdet });
        if (__float_as_uint(tr) != __float_as_uint(hr.hit ? hr.t : 0.f)) bad += 1;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// candidate lean reciprocals against the compiler's correctly rounded 1.f/x, every bit pattern;
// hist[v*256 + biased exponent of x] counts the mismatches of candidate v
extern "C" __global__ void rt_rcp_probe_kernel(unsigned long long *hist) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const uint32_t ex = ((uint32_t)b >> 23) & 255u;
        const uint32_t ref = __float_as_uint(1.f / x);
        const bool ref_nan = (ref & 0x7fffffffu) > 0x7f800000u;
        const float r0 = __builtin_amdgcn_rcpf(x);
        const float e0 = __builtin_fmaf(-x, r0, 1.f);
        const float r1 = __builtin_fmaf(e0, r0, r0);
        const float e1 = __builtin_fmaf(-x, r1, 1.f);
        const float r2 = __builtin_fmaf(e1, r1, r1);
        const float r2b = __builtin_fmaf(e1, r0, r1);       // the compiler's shape: residual times the first estimate
        const float cand[4] = { r0, r1, r2, r2b };
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const uint32_t a = __float_as_uint(cand[v]);
            const bool a_nan = (a & 0x7fffffffu) > 0x7f800000u;
            if (a != ref && !(a_nan && ref_nan)) atomicAdd(&hist[v * 256 + ex], 1ull);
        }
    }
}

// scalar building blocks, for rt_debug_eval
extern "C" __global__ void rt_eval_kernel(int op, const float *in, float *out, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float v = in[i], r = 0.f, t;
    switch (op) {
        case 0: dm_sincosf(v, r, t); break;
        case 1: dm_sincosf(v, t, r); break;
        case 2: r = dm_powf(v, 1.f / 2.2f); break;
        case 3: r = 1.f / v; break;
        case 4: r = sqrtf(v); break;
        case 5: r = (float)to_int(v); break;
        case 6: dm_sincosf_pos(v, r, t); break;
        case 7: dm_sincosf_pos(v, t, r); break;
        case 8: r = ieee_sqrt_lean(v); break;
        default: break;
    }
    out[i] = r;
}
#endif

}  // namespace RT_NS
}  // namespace rt
