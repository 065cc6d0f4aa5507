// rt_host.cpp -- host-side pieces either side of the render path, C ABI:
// the seed stream the reference draws from the C library, the camera basis, the built-in
// scene and the .scn reader.  Plain C++, strict binary32/binary64 (-ffp-contract=off).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/rt_api.h"

// rt_last_error()'s thread-local text lives in rt_api.hip; host helpers report through this
extern "C" void rt_host_set_error(const char *msg);

namespace {

int host_fail(const char *fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    rt_host_set_error(buf);
    return RT_ERR_ARG;
}

// Vec::norm, Vec.cpp:28-30: the squared length is binary32, the square root and the
// reciprocal are binary64 (unqualified sqrt on a float picks ::sqrt(double) there).
rt_vec3 host_norm(rt_vec3 v) {
    const float len2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float inv = static_cast<float>(1 / std::sqrt(static_cast<double>(len2)));
    return rt_vec3{ v.x * inv, v.y * inv, v.z * inv };
}

rt_vec3 host_cross(rt_vec3 a, rt_vec3 b) {
    return rt_vec3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}

}  // namespace

extern "C" {

void rt_compute_camera(rt_camera *cam, int w, int h) {
    if (!cam) return;
    cam->dir = host_norm(rt_vec3{ cam->target.x - cam->orig.x, cam->target.y - cam->orig.y,
                                  cam->target.z - cam->orig.z });
    const rt_vec3 up{ 0.f, 1.f, 0.f };
    const float fov = static_cast<float>((3.14159265358979323846 / 180.f) * 45.f);
    rt_vec3 cx = host_norm(host_cross(cam->dir, up));
    const float sx = w * fov / h;
    cam->x = rt_vec3{ cx.x * sx, cx.y * sx, cx.z * sx };
    rt_vec3 cy = host_norm(host_cross(cam->x, cam->dir));
    cam->y = rt_vec3{ cy.x * fov, cy.y * fov, cy.z * fov };
}

// glibc's rand() with its initial state (equivalent to srand(1)): the TYPE_3 additive
// generator x[i] = x[i-31] + x[i-3] (mod 2^32) over a 31-word table that is filled by the
// Lehmer generator 16807 * s mod (2^31 - 1), run for 310 steps before the first output;
// each output is x[i] >> 1.
// The stream is a pure function of the index, so its head is generated once per process and kept
// (up to kSeedCacheWords; a 1080p context needs 4.1 M words): creating a context, which
// rt_render() does on every call, then costs a copy instead of 4 million generator steps.
namespace {
struct RandState {
    uint32_t tab[31];
    int f, r;
    void init() {
        int64_t word = 1;
        tab[0] = 1;
        for (int i = 1; i < 31; ++i) {
            word = (16807 * word) % 2147483647;
            if (word < 0) word += 2147483647;
            tab[i] = static_cast<uint32_t>(word);
        }
        // front = index of x[i-31]'s slot, rear = x[i-3]'s; glibc starts them 3 apart
        f = 3;
        r = 0;
        for (int i = 0; i < 310; ++i) (void)step();
    }
    uint32_t step() {
        tab[f] += tab[r];
        const uint32_t v = tab[f] >> 1;
        f = (f + 1 == 31) ? 0 : f + 1;
        r = (r + 1 == 31) ? 0 : r + 1;
        return v;
    }
    uint32_t seed() {                                                 // OpenCLConfig.cpp:678-679
        const uint32_t v = step();
        return v < 2 ? 2u : v;
    }
};
constexpr size_t kSeedCacheWords = size_t(1) << 23;                   // 32 MiB
std::mutex g_seed_mutex;
std::vector<uint32_t> g_seed_head;                                    // seeds [0, size)
RandState g_seed_state;                                               // generator after g_seed_head.size() outputs
bool g_seed_started = false;
}  // namespace

void rt_default_seeds(uint32_t *seeds, size_t count) {
    if (!seeds) return;
    RandState tail;
    size_t have;
    {
        std::lock_guard<std::mutex> lock(g_seed_mutex);
        if (!g_seed_started) {
            g_seed_state.init();
            g_seed_started = true;
        }
        const size_t want = count < kSeedCacheWords ? count : kSeedCacheWords;
        if (g_seed_head.size() < want) {
            const size_t old = g_seed_head.size();
            g_seed_head.resize(want);
            for (size_t i = old; i < want; ++i) g_seed_head[i] = g_seed_state.seed();
        }
        have = count < g_seed_head.size() ? count : g_seed_head.size();
        memcpy(seeds, g_seed_head.data(), have * sizeof(uint32_t));
        tail = g_seed_state;                                          // continues at g_seed_head.size()
    }
    for (size_t i = have; i < count; ++i) seeds[i] = tail.seed();
}

int rt_demo_scene(rt_sphere *out, uint32_t cap) {
    static const rt_sphere demo[6] = {
        { 1000.f, { 0.f, -1000.f, 0.f }, { 0.f, 0.f, 0.f }, { 0.75f, 0.75f, 0.75f }, RT_DIFF },
        { 12.f, { 40.f, 20.f, 0.f }, { 0.f, 0.f, 0.f }, { 0.9f, 0.f, 0.f }, RT_REFR },
        { 11.f, { -35.f, 20.f, 0.f }, { 0.f, 0.f, 0.f }, { 0.f, 0.9f, 0.f }, RT_REFR },
        { 10.f, { 0.f, 25.f, -10.f }, { 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.9f }, RT_REFR },
        { 9.f, { 20.f, 10.f, -5.f }, { 0.f, 0.f, 0.f }, { 0.9f, 0.f, 0.9f }, RT_REFR },
        { 7.f, { 0.f, 60.f, 0.f }, { 12.f, 12.f, 12.f }, { 0.f, 0.f, 0.f }, RT_DIFF },
    };
    if (!out || cap < 6) return -6;
    memcpy(out, demo, sizeof demo);
    return 6;
}

int rt_read_scene(const char *path, rt_sphere *out, uint32_t cap, uint32_t *count, rt_vec3 *orig,
                  rt_vec3 *target, int reference_doubling) {
    if (!path || !out || !count || !orig || !target) return host_fail("rt_read_scene: null argument");
    *count = 0;
    FILE *f = fopen(path, "r");
    if (!f) return host_fail("Failed to open file: %s", path);
    int got = fscanf(f, "camera %f %f %f  %f %f %f\n", &orig->x, &orig->y, &orig->z, &target->x,
                     &target->y, &target->z);
    if (got != 6) {
        fclose(f);
        return host_fail("Failed to read 6 camera parameters: %d", got);
    }
    unsigned n = 0;
    got = fscanf(f, "size %u\n", &n);
    if (got != 1) {
        fclose(f);
        return host_fail("Failed to read sphere count: %d", got);
    }
    const uint64_t total = reference_doubling ? 2ull * n : n;
    if (total > cap) {
        fclose(f);
        return host_fail("scene has %llu spheres, capacity %u", (unsigned long long)total, cap);
    }
    uint32_t at = 0;
    if (reference_doubling) {            // Utility.cpp:120: vector built with n zeroed spheres...
        memset(out, 0, sizeof(rt_sphere) * n);
        at = n;
    }
    for (unsigned i = 0; i < n; ++i) {   // ...then n more are appended (:154)
        rt_sphere s;
        memset(&s, 0, sizeof s);
        int mat = 0;
        got = fscanf(f, "sphere %f  %f %f %f  %f %f %f  %f %f %f  %d\n", &s.rad, &s.p.x, &s.p.y, &s.p.z,
                     &s.e.x, &s.e.y, &s.e.z, &s.c.x, &s.c.y, &s.c.z, &mat);
        if (mat < 0 || mat > 2) {
            fclose(f);
            return host_fail("Failed to read material type for sphere #%u: %d", i, mat);
        }
        if (got != 11) {
            fclose(f);
            return host_fail("Failed to read sphere #%u: %d", i, got);
        }
        s.refl = mat;
        out[at++] = s;
    }
    fclose(f);
    *count = at;
    return RT_OK;
}

}  // extern "C"
