/*
 * rt_oracle.h -- CPU oracle for the per-pixel sphere path-trace hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the algorithm in
 * the reference's render kernel (SimpleRT/kernel/RayTracing_Kernel.cl) and of
 * the host code that feeds it (seed stream, camera basis).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product library (raytracing_simple_amd/csrc) never includes or links it.
 *
 * Parity status: PINNED against the reference itself: the reference's own kernel
 * source compiled in place as host C++ (oracle/_ref/libref.so, see oracle/Makefile)
 * and the fixtures generated from it (tests/golden (.npz files): the Demo scene and all
 * nine shipped .scn scenes).  SURVEY.md section 8c's camera hex, rand() outputs and
 * ray statistics reproduce; its FNV hashes do not.
 *
 * Arithmetic contract (what "the reference CPU path" means here):
 *   - IEEE-754 binary32 for every +,-,*,/ and sqrt, round-to-nearest-even,
 *     no fused contraction, denormals kept;
 *   - sin/cos/pow are glibc 2.35's sinf/cosf/powf (what the reference kernel's
 *     built-ins resolve to when its source is compiled as host C++), restated as
 *     a fixed sequence of IEEE binary64 +,-,*,fma (om_* below) that gives the
 *     same bits on any conforming machine;
 *   - the two RNG draws in light sampling are sequenced left to right.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Layouts shared bit-for-bit with the reference (Sphere.hpp:11-15, Camera.hpp:7-14,
 * RayTracing_Kernel.cl:6-8,38-45,51-55). */
typedef struct { float x, y, z; } orc_vec;                         /* 12 B */
typedef struct { float rad; orc_vec p, e, c; int32_t refl; } orc_sphere;   /* 44 B */
typedef struct { orc_vec orig, target, dir, x, y; } orc_camera;    /* 60 B */

enum { ORC_DIFF = 0, ORC_SPEC = 1, ORC_REFR = 2 };

/* Exact work counters. */
typedef struct {
    uint64_t samples;        /* camera rays = primary rays              */
    uint64_t closest_calls;  /* closest-hit queries (primary+extension) */
    uint64_t shadow_calls;   /* any-hit queries (shadow rays)           */
    uint64_t sphere_tests;   /* ray/sphere tests over both              */
    uint64_t rng_draws;
} orc_stats;

/* --- deterministic transcendental set: glibc 2.35 sinf/cosf/powf (FMA variant)
 *     restated as fixed binary64 arithmetic; see rt_oracle.c --- */
void  om_sincosf(float x, float *s, float *c);   /* |x| < 120 */
float om_sinf(float x);
float om_cosf(float x);
float om_powf(float x, float y);                 /* x >= 0 finite, |y*log2 x| < 126 */
float om_gammaf(float b);                        /* b^(1/2.2f), b in [0,1] */
uint64_t orc_math_mismatches(int which, uint32_t k0, uint32_t k1);

/* renderer math back-end: 0 = deterministic om_* set (default), 1 = host libm */
void orc_set_math_backend(int b);
int  orc_get_math_backend(void);

/* --- a3: the two-stream multiply-with-carry generator, .cl:143-169 --- */
float orc_get_random(uint32_t *s0, uint32_t *s1);

/* --- a13: seed stream = glibc rand() default stream, clamped to >= 2
 *     (OpenCLConfig.cpp:676-680; no srand anywhere in the reference). --- */
void orc_glibc_rand_stream(uint32_t *out, size_t n);
void orc_seeds_init(uint32_t *seeds, int w, int h);

/* --- a15: camera basis, Utility.cpp:71-85 + Vec.cpp:28-30 (double sqrt) --- */
void orc_camera_basis(orc_camera *cam, int w, int h);

/* --- a1: one pass (= one reference kernel launch) over rows [y0, y1). --- */
void orc_render_pass(orc_vec *colors, uint32_t *seeds, const orc_sphere *spheres,
                     uint32_t n_spheres, const orc_camera *cam, int w, int h,
                     int current_sample, uint32_t *pixels, int y0, int y1,
                     orc_stats *stats);

/* spp passes starting at sample `first_sample`; rows split over n_threads
 * (pixel streams are independent, so the split cannot change a bit). */
void orc_render(orc_vec *colors, uint32_t *seeds, const orc_sphere *spheres,
                uint32_t n_spheres, const orc_camera *cam, int w, int h,
                int first_sample, int spp, uint32_t *pixels, int n_threads,
                orc_stats *stats);

/* single-primitive probes used by unit tests */
float orc_sphere_intersect(const orc_sphere *s, const float o[3], const float d[3]);
void  orc_camera_ray(const orc_camera *cam, uint32_t *s0, uint32_t *s1, int w, int h,
                     int x, int y, float o[3], float d[3]);
int   orc_to_int(float v);

uint64_t orc_fnv1a64(const void *data, size_t n);

#ifdef __cplusplus
}
#endif
#endif
