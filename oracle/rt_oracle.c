/*
 * rt_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see rt_oracle.h for the contract).
 *
 * Restates, in plain C11, the algorithm of the reference render kernel
 *   /root/reference/SimpleRT/kernel/RayTracing_Kernel.cl   (cited below as ".cl:LINE")
 * and of the host code that prepares its inputs
 *   SimpleRT/src/OpenCLConfig.cpp:613-682  (seed stream)
 *   SimpleRT/src/Utility.cpp:71-85, SimpleRT/src/Vec.cpp:28-30  (camera basis).
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off (oracle/Makefile).  Contraction must
 * stay off: every product and sum below is a separately rounded binary32 operation,
 * written in the association order of the reference expression it restates.
 *
 * Parity status: PINNED.  tests/test_oracle_pins.py checks it against tests/golden (.npz files), the
 * outputs of the reference's own kernel source compiled as host C++ (oracle/_ref, built from
 * /root/reference by oracle/Makefile; fixtures made by tests/golden/make_golden.py) for the Demo
 * scene and all nine shipped .scn scenes, and directly against oracle/_ref where that build exists.
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* Deterministic transcendental set (om_*).                                    */
/*                                                                             */
/* The kernel's sin/cos/pow (.cl:34,209-210,404-405) are OpenCL built-ins, i.e.   */
/* a third-party dependency that is not in /root/reference.  When the kernel   */
/* source is compiled as host C++ (oracle/_ref) they resolve to glibc's        */
/* sinf/cosf/powf.  om_* restates that published algorithm -- glibc 2.35        */
/* (Ubuntu 2.35-0ubuntu3.11) sysdeps/ieee754/flt-32/{s_sinf.c,s_cosf.c,          */
/* s_sincosf.h,e_powf.c,e_powf_log2_data.c,e_exp2f_data.c}, itself ARM           */
/* optimized-routines' sinf/cosf/powf -- as evaluated by the x86-64 FMA          */
/* ifunc variant (__sinf_fma/__cosf_fma/__powf_fma): binary64 throughout, every */
/* multiply-add below that is written fma() is fused there, every other product */
/* and sum is rounded separately.  Coefficients and tables are the values of     */
/* __sincosf_table, __powf_log2_data and __exp2f_data in that libm.              */
/* Because the sequence is fixed IEEE binary64 arithmetic it gives the same bits */
/* on any conforming machine, CPU or GPU (tests: exhaustive equality with the    */
/* host libm over every argument the renderer can produce).                      */
/* Domains: sin/cos |x| < 120 (the renderer passes 2*pi*u, u in [0,1));          */
/*          pow x in [0, +inf) finite, |y*log2(x)| < 126.                        */
/* ------------------------------------------------------------------------- */
static double bits_to_f64(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static uint64_t f64_to_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static uint32_t f32_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float bits_to_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

#define OM_FMA(a, b, c) __builtin_fma((a), (b), (c))

/* __sincosf_table[0]: 2/pi * 2^24, pi/2, cosine c0..c4, sine s1..s3 */
#define OM_HPI_INV24 0x1.45f306dc9c883p+23
#define OM_HPI       0x1.921fb54442d18p+0
#define OM_C0 0x1.0000000000000p+0
#define OM_C1 -0x1.ffffffd0c621cp-2
#define OM_C2 0x1.55553e1068f19p-5
#define OM_C3 -0x1.6c087e89a359dp-10
#define OM_C4 0x1.99343027bf8c3p-16
#define OM_S1 -0x1.555545995a603p-3
#define OM_S2 0x1.1107605230bc4p-7
#define OM_S3 -0x1.994eb3774cf24p-13

static inline float om_sin_poly(double x, double x2)
{
    double x3 = x * x2;
    double s1 = OM_FMA(x2, OM_S3, OM_S2);
    double x7 = x3 * x2;
    double s = OM_FMA(x3, OM_S1, x);
    return (float)OM_FMA(s1, x7, s);
}

static inline double om_cos_poly_d(double x2)
{
    double x4 = x2 * x2;
    double c1 = OM_FMA(x2, OM_C1, OM_C0);
    double c2 = OM_FMA(x2, OM_C4, OM_C3);
    double x6 = x4 * x2;
    double c = OM_FMA(x4, OM_C2, c1);
    return OM_FMA(c2, x6, c);
}

/* quadrant n and reduced argument: n = round(x * 2/pi) through a 2^24-scaled
 * truncating conversion, xr = x - n*pi/2 with one fused rounding */
static inline double om_reduce(double x, int *n_out)
{
    double r = x * OM_HPI_INV24;
    int n = ((int32_t)r + 0x800000) >> 24;
    *n_out = n;
    return OM_FMA(-(double)n, OM_HPI, x);
}

/* sign[] of __sincosf_table: +,-,-,+ for n & 3 */
static inline double om_quadrant_sign(int n) { return ((n + 1) & 2) ? -1.0 : 1.0; }

void om_sincosf(float y, float *sin_out, float *cos_out)
{
    double x = (double)y;
    uint32_t top = (f32_to_bits(y) >> 20) & 0x7ff;
    if (top <= 0x3f3) {                       /* |y| < pi/4 */
        double x2 = x * x;
        if (top <= 0x397) {                   /* |y| < 2^-12 */
            *sin_out = y;
            *cos_out = 1.0f;
            return;
        }
        *sin_out = om_sin_poly(x, x2);
        *cos_out = (float)om_cos_poly_d(x2);
        return;
    }
    int n;
    double xr = om_reduce(x, &n);
    double x2 = xr * xr;
    float sp = om_sin_poly(xr * om_quadrant_sign(n), x2);
    double cd = om_cos_poly_d(x2);
    float cp = (float)((n & 2) ? -cd : cd);
    /* sinf: even n -> sine polynomial, odd n -> cosine polynomial; cosf uses n^1.
     * The sine polynomial's argument sign for cosf is sign[n & 3] as well. */
    if ((n & 1) == 0) {
        *sin_out = sp;
        *cos_out = cp;
    } else {
        *sin_out = cp;
        /* cosf, odd n: sinf_poly(xr * sign[n&3], x2, p, n^1) */
        *cos_out = sp;
    }
}

float om_sinf(float y) { float s, c; om_sincosf(y, &s, &c); return s; }
float om_cosf(float y) { float s, c; om_sincosf(y, &s, &c); return c; }

/* __powf_log2_data: 16 x {invc, logc}, polynomial A0..A4 (log2 scaled by 1) */
static const double om_log2_tab[16][2] = {
    { 0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2 }, { 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2 },
    { 0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2 }, { 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2 },
    { 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2 }, { 0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3 },
    { 0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3 }, { 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4 },
    { 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5 }, { 0x1.0000000000000p+0, 0x0.0p+0 },
    { 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4 },  { 0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3 },
    { 0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3 },  { 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2 },
    { 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2 },  { 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 },
};
#define OM_A0 0x1.27616c9496e0bp-2
#define OM_A1 -0x1.71969a075c67ap-2
#define OM_A2 0x1.ec70a6ca7baddp-2
#define OM_A3 -0x1.7154748bef6c8p-1
#define OM_A4 0x1.71547652ab82bp+0

/* __exp2f_data: tab[i] = bits(2^(i/32)) - (i << 47), shift = 0x1.8p52/32, C0..C2 */
static const uint64_t om_exp2_tab[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL,
};
#define OM_EXP2_SHIFT 0x1.8000000000000p+47
#define OM_E0 0x1.c6af84b912394p-5
#define OM_E1 0x1.ebfce50fac4f3p-3
#define OM_E2 0x1.62e42ff0c52d6p-1

float om_powf(float xf, float yf)
{
    uint32_t ix = f32_to_bits(xf);
    if ((ix << 1) == 0) return 0.f;                       /* pow(+-0, y > 0) */
    if (ix < 0x00800000u) {                                /* subnormal: renormalise */
        ix = f32_to_bits(xf * 0x1p23f) & 0x7fffffffu;
        ix -= 23u << 23;
    }
    /* log2(x) = k + log2(c_i) + log2(z/c_i), z in [0.7, 1.4) */
    uint32_t tmp = ix - 0x3f330000u;
    uint32_t i = (tmp >> 19) & 15u;
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = (int32_t)top >> 23;
    double z = (double)bits_to_f32(iz);
    double r = OM_FMA(z, om_log2_tab[i][0], -1.0);
    double y0 = om_log2_tab[i][1] + (double)k;
    double y = OM_FMA(r, OM_A0, OM_A1);
    double p = OM_FMA(r, OM_A2, OM_A3);
    double r2 = r * r;
    double q = OM_FMA(r, OM_A4, y0);
    double r4 = r2 * r2;
    q = OM_FMA(r2, p, q);
    double logx = OM_FMA(y, r4, q);

    double ylogx = (double)yf * logx;
    if (ylogx >= 126.0) ylogx = 126.0;                    /* outside the stated domain */
    if (ylogx <= -126.0) ylogx = -126.0;

    /* 2^ylogx = 2^(ki/32) * 2^rr, |rr| <= 1/64 */
    double kd = ylogx + OM_EXP2_SHIFT;
    uint64_t ki = f64_to_bits(kd);
    kd = kd - OM_EXP2_SHIFT;
    double rr = ylogx - kd;
    uint64_t t = om_exp2_tab[ki & 31u] + (ki << 47);
    double s = bits_to_f64(t);
    double zz = OM_FMA(rr, OM_E0, OM_E1);
    double rr2 = rr * rr;
    double w = OM_FMA(rr, OM_E2, 1.0);
    w = OM_FMA(zz, rr2, w);
    return (float)(w * s);
}

/* pow(b, 1.f/2.2f) of toInt, .cl:34; b already clamped to [0,1]. */
float om_gammaf(float b) { return om_powf(b, 1.f / 2.2f); }

/* exhaustive equality checks against the host libm (tests/test_detmath.py).
 * which: 0 = sinf/cosf over x = (2*pi)_f32 * k/2^23, k in [k0,k1);
 *        1 = powf(b, 1/2.2f) over every binary32 b with bits in [k0,k1). */
uint64_t orc_math_mismatches(int which, uint32_t k0, uint32_t k1)
{
    uint64_t bad = 0;
    if (which == 0) {
        for (uint32_t k = k0; k < k1; ++k) {
            float u = (float)k * 0x1p-23f;
            float x = (2.f * 3.14159265358979323846f) * u;
            float s, c;
            om_sincosf(x, &s, &c);
            if (f32_to_bits(s) != f32_to_bits(sinf(x))) ++bad;
            if (f32_to_bits(c) != f32_to_bits(cosf(x))) ++bad;
        }
    } else {
        const float e = 1.f / 2.2f;
        for (uint32_t k = k0; k < k1; ++k) {
            float b = bits_to_f32(k);
            if (f32_to_bits(om_powf(b, e)) != f32_to_bits(powf(b, e))) ++bad;
        }
    }
    return bad;
}

/* Math back-end used by the renderer below.
 *   0 (default)  the deterministic set above: what the HIP path is compared against;
 *   1            the host libm's sinf/cosf/powf: what oracle/_ref (the reference's kernel
 *                compiled as host C++) calls, so back-end 1 must reproduce _ref bit for
 *                bit -- that is the check that the restatement itself is exact.
 * On a host whose libm is glibc 2.35 with the FMA variants selected the two back-ends
 * are bit-identical (om_* restates exactly that algorithm). */
static int g_math_backend = 0;
void orc_set_math_backend(int b) { g_math_backend = b; }
int orc_get_math_backend(void) { return g_math_backend; }

static inline void rt_sincos(float x, float *s, float *c)
{
    if (g_math_backend == 1) { *s = sinf(x); *c = cosf(x); }
    else om_sincosf(x, s, c);
}

static inline float rt_gamma(float b)
{
    if (g_math_backend == 1) return powf(b, 1.f / 2.2f);
    return om_gammaf(b);
}

/* ------------------------------------------------------------------------- */
/* small binary32 vector helpers (.cl:72-138) -- by value, one rounding per op */
/* ------------------------------------------------------------------------- */
typedef orc_vec v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float k) { return v3_make(a.x * k, a.y * k, a.z * k); }
/* .cl:117-120: (x*x' + y*y') + z*z' */
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* .cl:122-126: f = 1/sqrt(v.v); v *= f */
static inline v3 v3_unit(v3 a) { float f = 1.f / sqrtf(v3_dot(a, a)); return v3_scale(a, f); }
/* .cl:128-131 */
static inline v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* .cl:135-138 -- the .y component is NOT examined (x is tested twice there). */
static inline int v3_is_dark(v3 a) { return (a.x == 0.f) && (a.z == 0.f); }

/* OpenCL sign(): +-1, +-0 kept, NaN -> 0 */
static inline float cl_sign(float x)
{
    if (x > 0.f) return 1.f;
    if (x < 0.f) return -1.f;
    if (x != x) return 0.f;
    return x;
}

#define RT_EPS 0.01f                      /* .cl:68 */
#define RT_PI  3.14159265358979323846f    /* .cl:69 */

typedef struct {
    uint32_t s0, s1;
    orc_stats *st;
} lane_rng;

/* ---- a3 (.cl:143-169) ---- */
float orc_get_random(uint32_t *s0, uint32_t *s1)
{
    uint32_t a = *s0, b = *s1;
    a = 36969u * (a & 65535u) + (a >> 16);
    b = 18000u * (b & 65535u) + (b >> 16);
    *s0 = a;
    *s1 = b;
    uint32_t word = (a << 16) + b;
    uint32_t fb = (word & 0x007fffffu) | 0x40000000u;    /* float in [2,4) */
    float f;
    memcpy(&f, &fb, 4);
    return (f - 2.f) / 2.f;
}

static inline float draw(lane_rng *g)
{
    g->st->rng_draws++;
    return orc_get_random(&g->s0, &g->s1);
}

/* ---- a8 (.cl:173-201) ---- */
static inline float hit_distance(const orc_sphere *sp, v3 o, v3 d)
{
    v3 op = v3_sub(sp->p, o);
    float b = v3_dot(op, d);
    float det = b * b - v3_dot(op, op) + sp->rad * sp->rad;
    if (det < 0.f) return 0.f;
    det = sqrtf(det);
    float t = b - det;
    if (t > RT_EPS) return t;
    t = b + det;
    if (t > RT_EPS) return t;
    return 0.f;
}

float orc_sphere_intersect(const orc_sphere *s, const float o[3], const float d[3])
{
    return hit_distance(s, v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]));
}

/* ---- a6 (.cl:215-232) ---- */
static int closest_hit(const orc_sphere *sph, uint32_t n, v3 o, v3 d, float *t_out,
                       uint32_t *id_out, orc_stats *st)
{
    float t = 1e20f;
    uint32_t id = 0;
    st->closest_calls++;
    st->sphere_tests += n;
    for (uint32_t i = 0; i < n; ++i) {
        float h = hit_distance(&sph[i], o, d);
        if (h != 0.f && h < t) { t = h; id = i; }
    }
    *t_out = t;
    *id_out = id;
    return t < 1e20f;
}

/* ---- a7 (.cl:234-247) ---- */
static int any_hit(const orc_sphere *sph, uint32_t n, v3 o, v3 d, float max_t, orc_stats *st)
{
    st->shadow_calls++;
    for (uint32_t i = 0; i < n; ++i) {
        st->sphere_tests++;
        float h = hit_distance(&sph[i], o, d);
        if (h != 0.f && h < max_t) return 1;
    }
    return 0;
}

/* ---- a9 + a10 (.cl:203-213, 249-303) ---- */
static v3 direct_light(const orc_sphere *sph, uint32_t n, lane_rng *g, v3 hp, v3 nl)
{
    v3 sum = v3_make(0.f, 0.f, 0.f);
    for (uint32_t i = 0; i < n; ++i) {
        const orc_sphere *L = &sph[i];
        if (v3_is_dark(L->e)) continue;

        /* two draws, first one is u1 (left-to-right, SURVEY hard part 2) */
        float u1 = draw(g);
        float u2 = draw(g);
        float zc = 1.f - 2.f * u1;
        float ring = sqrtf(fmaxf(0.f, 1.f - zc * zc));
        float phi = (2.f * RT_PI) * u2;
        float sphi, cphi;
        rt_sincos(phi, &sphi, &cphi);
        v3 unit = v3_make(ring * cphi, ring * sphi, zc);

        v3 on_light = v3_add(v3_scale(unit, L->rad), L->p);
        v3 sd = v3_sub(on_light, hp);
        float len = sqrtf(v3_dot(sd, sd));
        sd = v3_scale(sd, 1.f / len);

        float wo = v3_dot(sd, unit);
        if (wo > 0.f) continue;          /* far side of the light */
        wo = -wo;

        float wi = v3_dot(sd, nl);
        if (wi > 0.f && !any_hit(sph, n, hp, sd, len - RT_EPS, g->st)) {
            float k = (4.f * RT_PI * L->rad * L->rad) * wi * wo / (len * len);
            sum = v3_add(sum, v3_scale(L->e, k));
        }
    }
    return sum;
}

/* ---- a5 (.cl:305-491) ---- */
static v3 trace_path(const orc_sphere *sph, uint32_t n, v3 o, v3 d, lane_rng *g)
{
    v3 rad = v3_make(0.f, 0.f, 0.f);
    v3 thr = v3_make(1.f, 1.f, 1.f);
    int after_specular = 1;

    for (unsigned depth = 0; depth <= 7; ++depth) {       /* .cl:320: stop when depth > 7 */
        float t;
        uint32_t id;
        if (!closest_hit(sph, n, o, d, &t, &id, g->st)) return rad;
        const orc_sphere *obj = &sph[id];

        v3 hp = v3_add(o, v3_scale(d, t));
        v3 nrm = v3_unit(v3_sub(hp, obj->p));
        float dp = v3_dot(nrm, d);
        v3 nl = v3_scale(nrm, -1.f * cl_sign(dp));

        if (!v3_is_dark(obj->e)) {                          /* .cl:358-368 */
            if (after_specular) {
                v3 em = v3_mul(thr, v3_scale(obj->e, fabsf(dp)));
                rad = v3_add(rad, em);
            }
            return rad;
        }

        if (obj->refl == ORC_DIFF) {                        /* .cl:370-412 */
            after_specular = 0;
            thr = v3_mul(thr, obj->c);
            v3 ld = direct_light(sph, n, g, hp, nl);
            rad = v3_add(rad, v3_mul(thr, ld));

            float r1 = (2.f * RT_PI) * draw(g);
            float r2 = draw(g);
            float r2s = sqrtf(r2);
            v3 w = nl;
            v3 a = (fabsf(w.x) > .1f) ? v3_make(0.f, 1.f, 0.f) : v3_make(1.f, 0.f, 0.f);
            v3 u = v3_unit(v3_cross(a, w));
            v3 v = v3_cross(w, u);
            float s1, c1;
            rt_sincos(r1, &s1, &c1);
            v3 nd = v3_add(v3_scale(u, c1 * r2s), v3_scale(v, s1 * r2s));
            nd = v3_add(nd, v3_scale(w, sqrtf(1 - r2)));
            o = hp;
            d = nd;
        } else if (obj->refl == ORC_SPEC) {                 /* .cl:413-424 */
            after_specular = 1;
            v3 nd = v3_sub(d, v3_scale(nrm, 2.f * v3_dot(nrm, d)));
            thr = v3_mul(thr, obj->c);
            o = hp;
            d = nd;
        } else {                                            /* .cl:425-489 */
            after_specular = 1;
            v3 refl = v3_sub(d, v3_scale(nrm, 2.f * v3_dot(nrm, d)));
            int into = v3_dot(nrm, nl) > 0;
            const float nc = 1.f, nt = 1.52f;
            float nnt = into ? nc / nt : nt / nc;
            float ddn = v3_dot(d, nl);
            float cos2t = 1.f - nnt * nnt * (1.f - ddn * ddn);
            if (cos2t < 0.f) {                              /* total internal reflection */
                thr = v3_mul(thr, obj->c);
                o = hp;
                d = refl;
                continue;
            }
            float kk = (float)(into ? 1 : -1) * (ddn * nnt + sqrtf(cos2t));
            v3 td = v3_unit(v3_sub(v3_scale(d, nnt), v3_scale(nrm, kk)));
            float fa = nt - nc, fb = nt + nc;
            float R0 = fa * fa / (fb * fb);
            float c = 1 - (into ? -ddn : v3_dot(td, nrm));
            float Re = R0 + (1 - R0) * c * c * c * c * c;
            float Tr = 1.f - Re;
            float P = .25f + .5f * Re;
            float RP = Re / P;
            float TP = Tr / (1.f - P);
            if (draw(g) < P) {
                thr = v3_mul(v3_scale(thr, RP), obj->c);
                o = hp;
                d = refl;
            } else {
                thr = v3_mul(v3_scale(thr, TP), obj->c);
                o = hp;
                d = td;
            }
        }
    }
    return rad;
}

/* ---- a4 (.cl:494-549) ---- */
static void camera_ray(const orc_camera *cam, lane_rng *g, int w, int h, int x, int y,
                       v3 *o, v3 *d)
{
    float inv_w = 1.f / w;
    float inv_h = 1.f / h;
    float j1 = draw(g) - 0.5f;
    float j2 = draw(g) - 0.5f;
    float kcx = (x + j1) * inv_w - 0.5f;
    float kcy = (y + j2) * inv_h - 0.5f;
    v3 rd = v3_make(cam->x.x * kcx + cam->y.x * kcy + cam->dir.x,
                    cam->x.y * kcx + cam->y.y * kcy + cam->dir.y,
                    cam->x.z * kcx + cam->y.z * kcy + cam->dir.z);
    *o = v3_add(v3_scale(rd, 0.1f), cam->orig);
    *d = v3_unit(rd);
}

void orc_camera_ray(const orc_camera *cam, uint32_t *s0, uint32_t *s1, int w, int h, int x,
                    int y, float o[3], float d[3])
{
    orc_stats st = { 0 };
    lane_rng g = { *s0, *s1, &st };
    v3 ro, rd;
    camera_ray(cam, &g, w, h, x, y, &ro, &rd);
    *s0 = g.s0; *s1 = g.s1;
    o[0] = ro.x; o[1] = ro.y; o[2] = ro.z;
    d[0] = rd.x; d[1] = rd.y; d[2] = rd.z;
}

/* ---- a2 (.cl:34) ---- */
int orc_to_int(float v)
{
    float cl = fminf(fmaxf(v, 0.f), 1.f);
    return (int)(rt_gamma(cl) * 255.f + .5f);
}

/* ---- a1 (.cl:551-600): one sample for one pixel ---- */
static inline void pixel_pass(orc_vec *colors, uint32_t *seeds, const orc_sphere *sph,
                              uint32_t n, const orc_camera *cam, int w, int h, int sample,
                              uint32_t *pixels, int x, int y, orc_stats *st)
{
    size_t gid = (size_t)y * (size_t)w + (size_t)x;
    lane_rng g = { seeds[2 * gid], seeds[2 * gid + 1], st };
    v3 o, d;
    st->samples++;
    camera_ray(cam, &g, w, h, x, y, &o, &d);
    v3 r = trace_path(sph, n, o, d, &g);

    size_t ci = (size_t)(h - y - 1) * (size_t)w + (size_t)x;   /* colour plane is y-flipped */
    if (sample == 0) {
        colors[ci] = r;
    } else {
        float k1 = (float)sample;
        float k2 = 1.f / (sample + 1.f);
        colors[ci].x = (colors[ci].x * k1 + r.x) * k2;
        colors[ci].y = (colors[ci].y * k1 + r.y) * k2;
        colors[ci].z = (colors[ci].z * k1 + r.z) * k2;
    }
    pixels[gid] = (uint32_t)(orc_to_int(colors[ci].x) | (orc_to_int(colors[ci].y) << 8) |
                             (orc_to_int(colors[ci].z) << 16));
    seeds[2 * gid] = g.s0;
    seeds[2 * gid + 1] = g.s1;
}

void orc_render_pass(orc_vec *colors, uint32_t *seeds, const orc_sphere *spheres,
                     uint32_t n_spheres, const orc_camera *cam, int w, int h,
                     int current_sample, uint32_t *pixels, int y0, int y1, orc_stats *stats)
{
    orc_stats local = { 0 };
    for (int y = y0; y < y1; ++y)
        for (int x = 0; x < w; ++x)
            pixel_pass(colors, seeds, spheres, n_spheres, cam, w, h, current_sample, pixels, x,
                       y, &local);
    if (stats) {
        stats->samples += local.samples;
        stats->closest_calls += local.closest_calls;
        stats->shadow_calls += local.shadow_calls;
        stats->sphere_tests += local.sphere_tests;
        stats->rng_draws += local.rng_draws;
    }
}

typedef struct {
    orc_vec *colors; uint32_t *seeds; const orc_sphere *sph; uint32_t n;
    const orc_camera *cam; int w, h, first, spp; uint32_t *pixels; int y0, y1;
    orc_stats st;
} band_job;

static void *band_main(void *arg)
{
    band_job *j = (band_job *)arg;
    memset(&j->st, 0, sizeof j->st);
    /* pass-major inside the band, as the reference's host loop does per launch */
    for (int s = j->first; s < j->first + j->spp; ++s)
        orc_render_pass(j->colors, j->seeds, j->sph, j->n, j->cam, j->w, j->h, s, j->pixels,
                        j->y0, j->y1, &j->st);
    return NULL;
}

typedef struct { band_job *jobs; int n_bands, tid, stride; } band_worker;

static void *worker_main(void *arg)
{
    band_worker *w = (band_worker *)arg;
    for (int b = w->tid; b < w->n_bands; b += w->stride) band_main(&w->jobs[b]);
    return NULL;
}

void orc_render(orc_vec *colors, uint32_t *seeds, const orc_sphere *spheres,
                uint32_t n_spheres, const orc_camera *cam, int w, int h, int first_sample,
                int spp, uint32_t *pixels, int n_threads, orc_stats *stats)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > h) n_threads = h > 0 ? h : 1;
    /* interleave small row bands over threads so open scenes balance */
    enum { BAND = 4 };
    int n_bands = (h + BAND - 1) / BAND;
    band_job *jobs = (band_job *)calloc((size_t)n_bands, sizeof *jobs);
    for (int b = 0; b < n_bands; ++b) {
        band_job *j = &jobs[b];
        j->colors = colors; j->seeds = seeds; j->sph = spheres; j->n = n_spheres;
        j->cam = cam; j->w = w; j->h = h; j->first = first_sample; j->spp = spp;
        j->pixels = pixels; j->y0 = b * BAND; j->y1 = (b + 1) * BAND < h ? (b + 1) * BAND : h;
    }
    if (n_threads == 1) {
        for (int b = 0; b < n_bands; ++b) band_main(&jobs[b]);
    } else {
        pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof *th);
        band_worker *ws = (band_worker *)calloc((size_t)n_threads, sizeof *ws);
        for (int t = 0; t < n_threads; ++t) {
            ws[t].jobs = jobs; ws[t].n_bands = n_bands; ws[t].tid = t; ws[t].stride = n_threads;
            pthread_create(&th[t], NULL, worker_main, &ws[t]);
        }
        for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
        free(th);
        free(ws);
    }
    if (stats) {
        for (int b = 0; b < n_bands; ++b) {
            stats->samples += jobs[b].st.samples;
            stats->closest_calls += jobs[b].st.closest_calls;
            stats->shadow_calls += jobs[b].st.shadow_calls;
            stats->sphere_tests += jobs[b].st.sphere_tests;
            stats->rng_draws += jobs[b].st.rng_draws;
        }
    }
    free(jobs);
}

/* ---- a13: glibc rand() default stream (TYPE_3 additive feedback, seed 1) ---- */
void orc_glibc_rand_stream(uint32_t *out, size_t n)
{
    int32_t init[34];
    init[0] = 1;
    for (int i = 1; i < 31; ++i) {
        int64_t v = (16807LL * init[i - 1]) % 2147483647LL;
        if (v < 0) v += 2147483647LL;
        init[i] = (int32_t)v;
    }
    for (int i = 31; i < 34; ++i) init[i] = init[i - 31];

    uint32_t ring[34];
    for (int i = 0; i < 34; ++i) ring[i] = (uint32_t)init[i];
    /* element i (i >= 34) = element[i-31] + element[i-3]; the first 310 are discarded */
    size_t idx = 34;
    for (size_t i = 0; i < 310 + n; ++i, ++idx) {
        uint32_t v = ring[(idx - 31) % 34] + ring[(idx - 3) % 34];
        ring[idx % 34] = v;
        if (i >= 310) out[i - 310] = v >> 1;
    }
}

void orc_seeds_init(uint32_t *seeds, int w, int h)
{
    size_t n = 2 * (size_t)w * (size_t)h;
    orc_glibc_rand_stream(seeds, n);
    for (size_t i = 0; i < n; ++i)
        if (seeds[i] < 2) seeds[i] = 2;            /* OpenCLConfig.cpp:678-679 */
}

/* ---- a15: Vec::norm uses the double sqrt and a double reciprocal (Vec.cpp:28-30) ---- */
static v3 host_norm(v3 a)
{
    float s = a.x * a.x + a.y * a.y + a.z * a.z;
    float f = (float)(1 / sqrt((double)s));
    return v3_scale(a, f);
}

void orc_camera_basis(orc_camera *cam, int w, int h)
{
    cam->dir = host_norm(v3_sub(cam->target, cam->orig));
    v3 up = v3_make(0.f, 1.f, 0.f);
    const float fov = (float)((3.14159265358979323846 / 180.f) * 45.f);
    cam->x = host_norm(v3_cross(cam->dir, up));
    cam->x = v3_scale(cam->x, w * fov / h);
    cam->y = host_norm(v3_cross(cam->x, cam->dir));
    cam->y = v3_scale(cam->y, fov);
}

uint64_t orc_fnv1a64(const void *data, size_t n)
{
    const unsigned char *p = (const unsigned char *)data;
    uint64_t hsh = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < n; ++i) { hsh ^= p[i]; hsh *= 0x100000001b3ULL; }
    return hsh;
}
