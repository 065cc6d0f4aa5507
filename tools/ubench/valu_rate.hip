// valu_rate.hip -- measured VALU issue rates on gfx950 (wave64), used to price the render kernel.
// Each kernel runs a long unrolled stream of ONE instruction kind on independent registers;
// grid = 256 CUs x 4 SIMDs x W waves.  Reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, db = 1.0001, dc = 0.5;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // v_mul_f32
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 1) {  // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 2) {  // v_pk_mul_f32 on register pairs
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b};
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                              "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));)
            a0 = p0.x + p0.y; a2 = p1.x + p1.y; a4 = p2.x + p2.y; a6 = p3.x + p3.y;
        } else if (KIND == 3) {  // v_sqrt_f32
            REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                              "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 4) {  // v_fma_f64
            REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                              "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));)
        } else if (KIND == 5) {  // v_cmp + v_cndmask pairs (two instructions per pair)
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %4, %4, %5, vcc\n"
                              "v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %0, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %6, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
        } else if (KIND == 6) {  // v_mul_f64
            REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                              "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));)
        } else if (KIND == 7) {  // v_mul_u32_u24 / integer
            int i0 = __float_as_int(a0), i1 = __float_as_int(a1), i2 = __float_as_int(a2), i3 = __float_as_int(a3);
            REP8(asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0\n"
                              "v_lshrrev_b32 %0, 3, %0\n v_and_b32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_lshlrev_b32 %3, 1, %3\n"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(12345));)
            a0 = __int_as_float(i0 ^ i1 ^ i2 ^ i3);
        } else if (KIND == 8) {  // v_pk_fma_f32
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
            a0 = p0.x + p0.y; a2 = p1.x + p1.y; a4 = p2.x + p2.y; a6 = p3.x + p3.y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3);
}

template <int KIND>
void run(const char *name, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
    float *out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_wave = (double)iters * 64;
    const double ns_per_inst_per_simd = ms * 1e6 / (insts_per_wave * waves_per_simd);
    printf("%-14s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n", name, waves_per_simd, ms,
           ns_per_inst_per_simd, ns_per_inst_per_simd * 2.4);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_mul_f32", w);
        run<1>("v_fma_f32", w);
        run<2>("v_pk_mul_f32", w);
        run<8>("v_pk_fma_f32", w);
        run<3>("v_sqrt_f32", w);
        run<4>("v_fma_f64", w);
        run<6>("v_mul_f64", w);
        run<5>("cmp+cndmask", w);
        run<7>("int mix", w);
    }
    return 0;
}
