// rt_inflight.cpp -- bench.py's single-GPU frame loop as a native host program: F contexts, each with
// its own stream (rt_stream), K independent frames of the C2 workload kept F in flight through the C ABI
// only (rt_reset_async + rt_render_async), timed with the host clock around a device synchronisation.
//   hipcc -O2 -std=c++17 -Iinclude tools/rt_inflight.cpp -o raytracing_simple_amd/rt_inflight \
//         -Lraytracing_simple_amd -lrt_hip -Wl,-rpath,'$ORIGIN'
//   raytracing_simple_amd/rt_inflight [F] [frames] [w h spp] [fast]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_api.h"

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 2, frames = argc > 2 ? atoi(argv[2]) : 40;
    const int w = argc > 5 ? atoi(argv[3]) : 1920, h = argc > 5 ? atoi(argv[4]) : 1080, spp = argc > 5 ? atoi(argv[5]) : 64;
    const bool fast = argc > 6 && !strcmp(argv[6], "fast");
    const int raw_mode = -1;
    rt_sphere sph[6];
    if (rt_demo_scene(sph, 6) != 6) return 1;
    rt_camera cam{};
    cam.orig = rt_vec3{ 20.f, 100.f, 120.f };
    cam.target = rt_vec3{ 0.f, 25.f, 0.f };
    rt_compute_camera(&cam, w, h);
    std::vector<rt_ctx*> ctx(F, nullptr);
    for (auto& c : ctx) {
        if (rt_create(&c, w, h) != RT_OK || rt_set_scene(c, sph, 6) != RT_OK || rt_set_camera(c, &cam) != RT_OK ||
            rt_set_mode(c, raw_mode >= 0 ? raw_mode : (fast ? RT_MODE_FAST : RT_MODE_PARITY)) != RT_OK) {
            fprintf(stderr, "setup failed: %s\n", rt_last_error());
            return 1;
        }
    }
    auto frame = [&](int k) {
        rt_ctx* c = ctx[k % F];
        return rt_reset_async(c, rt_stream(c)) == RT_OK && rt_render_async(c, spp, rt_stream(c)) == RT_OK;
    };
    for (int k = 0; k < 2 * F; ++k)
        if (!frame(k)) return 1;                                  // warm-up
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < frames; ++k)
        if (!frame(k)) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // exact ray count of one frame, and the frame itself for a checksum
    std::vector<uint32_t> px((size_t)w * h);
    rt_stats st{};
    if (rt_reset(ctx[0]) != RT_OK || rt_render_pass(ctx[0], px.data(), spp) != RT_OK || rt_get_stats(ctx[0], &st) != RT_OK) return 1;
    unsigned long long sum = 0;
    for (uint32_t v : px) sum = sum * 1099511628211ull + v;
    const double rays = (double)(st.samples + st.shadow_rays);
    printf("{\"frames_in_flight\": %d, \"frames\": %d, \"w\": %d, \"h\": %d, \"spp\": %d, \"mode\": \"%s\", \"ms_per_frame\": %.4f, "
           "\"Mray_s_primary_shadow\": %.1f, \"frame_checksum\": \"%016llx\"}\n",
           F, frames, w, h, spp, raw_mode >= 0 ? argv[6] : (fast ? "fast" : "parity"), ms / frames, rays * frames / (ms * 1e3), sum);
    for (auto c : ctx) rt_destroy(c);
    return 0;
}
