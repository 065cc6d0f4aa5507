// ref_shim.cpp -- builds oracle/_ref/libref.so: the REFERENCE's own kernel source and
// host sources, compiled where they lie under /root/reference (path given by
// -DREF_ROOT=...), nothing copied.  TEST INFRASTRUCTURE: used only to validate the CPU
// restatement (rt_oracle.c) and to generate tests/golden fixtures in this container.
// The reference never travels to the GPU box; the built .so is git-ignored.
//
// The kernel file is OpenCL C.  What OpenCL C provides as part of the LANGUAGE (address
// space qualifiers, get_global_id, clamp/max/min/sign on float) is given here with the
// OpenCL 2.0 specification's semantics (6.13.4 common functions); the math calls resolve
// to the host's <cmath> binary32 overloads.  No header, library or generated file of the
// reference is replaced.  Compile with clang++ -ffp-contract=off: clang sequences the two
// draws at .cl:275 left to right, like the clang-based OpenCL compilers the kernel targets.
#include <cmath>
#include <math.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "Camera.hpp"
#include "Scene.hpp"
#include "Sphere.hpp"
#include "Utility.hpp"

#define RS_STR2(x) #x
#define RS_STR(x) RS_STR2(x)

namespace k {
#define __kernel
#define __global
#define __constant
static thread_local int g_gid;
static inline int get_global_id(int) { return g_gid; }
static inline float clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float max(float a, float b) { return a < b ? b : a; }
static inline float min(float a, float b) { return b < a ? b : a; }
static inline float sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : (x != x ? 0.f : x)); }
#include RS_STR(REF_ROOT/SimpleRT/kernel/RayTracing_Kernel.cl)
}  // namespace k

static_assert(sizeof(Sphere) == 44 && sizeof(k::Sphere) == 44, "sphere layout");
static_assert(sizeof(Camera) == 60 && sizeof(k::Camera) == 60, "camera layout");

extern "C" {

// seeds exactly as OpenCLConfigBuffer::allocateBuffer fills them (OpenCLConfig.cpp:676-680);
// srand(1) is the documented equivalent of "never seeded".
void ref_seeds_init(unsigned* seeds, int w, int h) {
    std::srand(1);
    for (long i = 0; i < 2L * w * h; ++i) {
        seeds[i] = std::rand();
        if (seeds[i] < 2) seeds[i] = 2;
    }
}

void ref_camera_basis(float* cam15, int w, int h) {
    Camera c;
    std::memcpy(&c, cam15, sizeof c);
    computeCameraVariables(&c, w, h);
    std::memcpy(cam15, &c, sizeof c);
}

int ref_demo_scene(void* spheres44, int cap) {
    int n = (int)DemoSpheres.size();
    if (n > cap) return -n;
    std::memcpy(spheres44, DemoSpheres.data(), (size_t)n * sizeof(Sphere));
    return n;
}

// readScene as shipped, including its doubling of the vector (Utility.cpp:120,154)
int ref_read_scene(const char* path, void* spheres44, int cap, float* orig3, float* target3) {
    Vec o, t;
    std::vector<Sphere> s = readScene(path, o, t);
    int n = (int)s.size();
    orig3[0] = o.x; orig3[1] = o.y; orig3[2] = o.z;
    target3[0] = t.x; target3[1] = t.y; target3[2] = t.z;
    if (n > cap) return -n;
    std::memcpy(spheres44, s.data(), (size_t)n * sizeof(Sphere));
    return n;
}

// one reference kernel launch: every work-item of the 1-D range, in gid order
void ref_render_pass(float* colors, unsigned* seeds, const void* spheres44, unsigned n,
                     const float* cam15, int w, int h, int current_sample, int* pixels) {
    for (int gid = 0; gid < w * h; ++gid) {
        k::g_gid = gid;
        k::RayTracing((k::Vec*)colors, seeds, (k::Sphere*)spheres44, (k::Camera*)cam15, n, w, h,
                      current_sample, pixels);
    }
}

// `n_passes` launches on `threads` host threads.  Work-items are independent (own seed pair, own
// colour and pixel slot), so each thread runs all passes over its slice of the 1-D range; the
// buffers end up exactly as after n_passes calls of ref_render_pass.  Used by bench.py's
// cpu_baseline leg (kind "reference") on the GPU box's host cores.
void ref_render_passes_mt(float* colors, unsigned* seeds, const void* spheres44, unsigned n,
                          const float* cam15, int w, int h, int first_sample, int n_passes,
                          int* pixels, int threads) {
    if (threads < 1) threads = 1;
    const long total = (long)w * h;
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) {
        const long lo = total * t / threads, hi = total * (t + 1) / threads;
        pool.emplace_back([=]() {
            for (int s = first_sample; s < first_sample + n_passes; ++s)
                for (long gid = lo; gid < hi; ++gid) {
                    k::g_gid = (int)gid;
                    k::RayTracing((k::Vec*)colors, seeds, (k::Sphere*)spheres44, (k::Camera*)cam15, n, w, h, s, pixels);
                }
        });
    }
    for (auto& th : pool) th.join();
}

float ref_get_random(unsigned* s0, unsigned* s1) { return k::GetRandom(s0, s1); }

float ref_sphere_intersect(const void* sphere44, const float* o, const float* d) {
    k::Ray r;
    r.o.x = o[0]; r.o.y = o[1]; r.o.z = o[2];
    r.d.x = d[0]; r.d.y = d[1]; r.d.z = d[2];
    return k::SphereIntersect((const k::Sphere*)sphere44, &r);
}

}  // extern "C"
