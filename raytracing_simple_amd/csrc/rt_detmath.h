// rt_detmath.h -- device transcendental set of the render kernels.
//
// Parity set (dm_*): the kernel's sin/cos/pow built-ins (RayTracing_Kernel.cl:34,209-210,
// 404-405) as glibc 2.35's sinf/cosf/powf evaluate them (x86-64 FMA variant): binary64
// arithmetic, a fixed sequence of mul / add / fma, so v_mul_f64 / v_add_f64 / v_fma_f64
// reproduce the host's bits.  Algorithm and coefficients: glibc 2.35
// sysdeps/ieee754/flt-32/{s_sincosf.h,e_powf.c,e_powf_log2_data.c,e_exp2f_data.c} (ARM
// optimized-routines sinf/cosf/powf).  Built with -ffp-contract=off: only the fma()s written
// here are fused.
// Fast set (fm_*): v_sin_f32 / v_cos_f32 / v_log_f32 / v_exp_f32.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

#define RT_DEV __device__ __forceinline__

RT_DEV double dm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// __sincosf_table[0]
#define DM_HPI_INV24 0x1.45f306dc9c883p+23
#define DM_HPI       0x1.921fb54442d18p+0
#define DM_C0 0x1.0000000000000p+0
#define DM_C1 -0x1.ffffffd0c621cp-2
#define DM_C2 0x1.55553e1068f19p-5
#define DM_C3 -0x1.6c087e89a359dp-10
#define DM_C4 0x1.99343027bf8c3p-16
#define DM_S1 -0x1.555545995a603p-3
#define DM_S2 0x1.1107605230bc4p-7
#define DM_S3 -0x1.994eb3774cf24p-13

RT_DEV float dm_sin_poly(double x, double x2) {
    double x3 = x * x2;
    double s1 = dm_fma(x2, DM_S3, DM_S2);
    double x7 = x3 * x2;
    double s = dm_fma(x3, DM_S1, x);
    return (float)dm_fma(s1, x7, s);
}

RT_DEV double dm_cos_poly(double x2) {
    double x4 = x2 * x2;
    double c1 = dm_fma(x2, DM_C1, DM_C0);
    double c2 = dm_fma(x2, DM_C4, DM_C3);
    double x6 = x4 * x2;
    double c = dm_fma(x4, DM_C2, c1);
    return dm_fma(c2, x6, c);
}

// sinf(y) and cosf(y), |y| < 120
RT_DEV void dm_sincosf(float y, float &sn, float &cs) {
    double x = (double)y;
    uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;
    if (top <= 0x3f3u) {                    // |y| < pi/4
        if (top <= 0x397u) {                // |y| < 2^-12
            sn = y;
            cs = 1.0f;
            return;
        }
        double x2 = x * x;
        sn = dm_sin_poly(x, x2);
        cs = (float)dm_cos_poly(x2);
        return;
    }
    double r = x * DM_HPI_INV24;
    int n = ((int)r + 0x800000) >> 24;      // round(x * 2/pi)
    double xr = dm_fma(-(double)n, DM_HPI, x);
    double x2 = xr * xr;
    double sgn = ((n + 1) & 2) ? -1.0 : 1.0;            // sign[n & 3] = +,-,-,+
    float sp = dm_sin_poly(xr * sgn, x2);
    double cd = dm_cos_poly(x2);
    float cp = (float)((n & 2) ? -cd : cd);
    bool odd = (n & 1) != 0;
    sn = odd ? cp : sp;
    cs = odd ? sp : cp;
}

// The same function without the two small-argument shortcuts, for y = +0 or y in [2^-149, 120):
// with n = 0 the reduction returns y itself, so the general path already evaluates the
// small-argument polynomial, and below 2^-12 the polynomial value rounds to y (sin) and to 1
// (cos).  Bit-identical to dm_sincosf over every argument the renderer forms (2*pi*k/2^23;
// tests/test_gpu_parity.py) and free of wave divergence.
RT_DEV void dm_sincosf_pos(float y, float &sn, float &cs) {
    double x = (double)y;
    double r = x * DM_HPI_INV24;
    int n = ((int)r + 0x800000) >> 24;
    double xr = dm_fma(-(double)n, DM_HPI, x);
    double x2 = xr * xr;
    double sgn = ((n + 1) & 2) ? -1.0 : 1.0;
    float sp = dm_sin_poly(xr * sgn, x2);
    double cd = dm_cos_poly(x2);
    float cp = (float)((n & 2) ? -cd : cd);
    bool odd = (n & 1) != 0;
    sn = odd ? cp : sp;
    cs = odd ? sp : cp;
}

// __powf_log2_data / __exp2f_data
static __device__ __constant__ const double dm_log2_tab[16][2] = {
    { 0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2 }, { 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2 },
    { 0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2 }, { 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2 },
    { 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2 }, { 0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3 },
    { 0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3 }, { 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4 },
    { 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5 }, { 0x1.0000000000000p+0, 0x0.0p+0 },
    { 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4 },  { 0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3 },
    { 0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3 },  { 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2 },
    { 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2 },  { 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 },
};
static __device__ __constant__ const uint64_t dm_exp2_tab[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL,
};
#define DM_A0 0x1.27616c9496e0bp-2
#define DM_A1 -0x1.71969a075c67ap-2
#define DM_A2 0x1.ec70a6ca7baddp-2
#define DM_A3 -0x1.7154748bef6c8p-1
#define DM_A4 0x1.71547652ab82bp+0
#define DM_EXP2_SHIFT 0x1.8000000000000p+47
#define DM_E0 0x1.c6af84b912394p-5
#define DM_E1 0x1.ebfce50fac4f3p-3
#define DM_E2 0x1.62e42ff0c52d6p-1

// powf(x, y) for finite x >= 0 and |y*log2 x| < 126
RT_DEV float dm_powf(float xf, float yf) {
    uint32_t ix = __float_as_uint(xf);
    if ((ix << 1) == 0u) return 0.f;
    if (ix < 0x00800000u) {                                // subnormal x
        ix = __float_as_uint(xf * 0x1p23f) & 0x7fffffffu;
        ix -= 23u << 23;
    }
    uint32_t tmp = ix - 0x3f330000u;
    uint32_t i = (tmp >> 19) & 15u;
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = (int)top >> 23;
    double z = (double)__uint_as_float(iz);
    double r = dm_fma(z, dm_log2_tab[i][0], -1.0);
    double y0 = dm_log2_tab[i][1] + (double)k;
    double y = dm_fma(r, DM_A0, DM_A1);
    double p = dm_fma(r, DM_A2, DM_A3);
    double r2 = r * r;
    double q = dm_fma(r, DM_A4, y0);
    double r4 = r2 * r2;
    q = dm_fma(r2, p, q);
    double logx = dm_fma(y, r4, q);

    double ylogx = (double)yf * logx;
    if (ylogx >= 126.0) ylogx = 126.0;
    if (ylogx <= -126.0) ylogx = -126.0;

    double kd = ylogx + DM_EXP2_SHIFT;
    uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd = kd - DM_EXP2_SHIFT;
    double rr = ylogx - kd;
    uint64_t t = dm_exp2_tab[ki & 31u] + (ki << 47);
    double s = __longlong_as_double((long long)t);
    double zz = dm_fma(rr, DM_E0, DM_E1);
    double rr2 = rr * rr;
    double w = dm_fma(rr, DM_E2, 1.0);
    w = dm_fma(zz, rr2, w);
    return (float)(w * s);
}

// ---- fast set -------------------------------------------------------------------------
// v_sin_f32 / v_cos_f32 take revolutions: sin(2*pi*u) = v_sin(u)
RT_DEV void fm_sincos_turns(float u, float &sn, float &cs) {
    sn = __builtin_amdgcn_sinf(u);
    cs = __builtin_amdgcn_cosf(u);
}
RT_DEV float fm_powf(float x, float y) {       // x >= 0
    return x > 0.f ? __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)) : 0.f;
}

}  // namespace rt
