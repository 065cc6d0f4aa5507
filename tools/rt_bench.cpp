// rt_bench.cpp -- headless C++ host for the HIP render path: what the reference's Main.cpp
// does (SimpleRT/src/Main.cpp:18-113) minus the GLUT window, with an image writer instead.
//
//   rt_bench <framework ID> <CPU/GPU (0/1)> <mem (0/1/2)> [scene.scn]
//            [--w W] [--h H] [--spp N] [--passes-per-launch K] [--pin] [--readback-ms T] [--mode parity|fast]
//            [--no-doubling] [--out frame.ppm] [--oneshot K] [--gpus N]
//   --oneshot K   render through the headline call rt_render(scene, cam, out, w, h, spp) K times instead of a
//                 context (prints the wall time of every call: the first builds the device state, the rest reuse it)
//   --gpus N      a multi-device context (rt_create_multi: N GPUs of this process, one RCCL gather per frame)
//   --rehearse N  the same with all N shards on device 0 (rt_create_multi_on: the one-GPU rehearsal of that path)
//
// The four positional arguments are the reference's; only framework ID 2 (the slot
// Config.cpp:63-65 leaves empty) is served, GPU = 1, memory type 0 (Buffer).
// Prints one JSON line with frame time and ray throughput.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_api.h"

static int die(const char* what) {
    fprintf(stderr, "%s: %s\n", what, rt_last_error());
    return 1;
}

static bool write_ppm(const std::string& path, const std::vector<uint32_t>& px, int w, int h) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    std::vector<unsigned char> row(static_cast<size_t>(w) * 3);
    for (int y = h - 1; y >= 0; --y) {          // buffer row 0 is the bottom of the image
        for (int x = 0; x < w; ++x) {
            uint32_t p = px[static_cast<size_t>(y) * w + x];
            row[3 * x] = p & 255;
            row[3 * x + 1] = (p >> 8) & 255;
            row[3 * x + 2] = (p >> 16) & 255;
        }
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
    return true;
}

int main(int argc, char** argv) {
    int w = 800, h = 600, spp = 1, per_launch = 0, mode = RT_MODE_PARITY, oneshot = 0, gpus = 1, rehearse = 0;
    bool pin = false;
    double readback_ms = 0.0;   // > 0: copy the frame out only when the last copy is this old (the adapter's display cadence)   // SetupGL.cpp:32-33
    bool doubling = true;
    std::string out, scene_path;
    std::vector<const char*> pos;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { return (i + 1 < argc) ? argv[++i] : ""; };
        if (a == "--w") w = atoi(next());
        else if (a == "--h") h = atoi(next());
        else if (a == "--spp") spp = atoi(next());
        else if (a == "--passes-per-launch") per_launch = atoi(next());
        else if (a == "--pin") pin = true;
        else if (a == "--readback-ms") readback_ms = atof(next());
        else if (a == "--mode") mode = strcmp(next(), "fast") == 0 ? RT_MODE_FAST : RT_MODE_PARITY;
        else if (a == "--no-doubling") doubling = false;
        else if (a == "--oneshot") oneshot = atoi(next());
        else if (a == "--gpus") gpus = atoi(next());
        else if (a == "--rehearse") rehearse = atoi(next());
        else if (a == "--out") out = next();
        else pos.push_back(argv[i]);
    }
    if (!pos.empty() && atoi(pos[0]) != 2) {
        fprintf(stderr, "Unsupported Framework Type (this host serves framework ID 2 = HIP)\n");
        return 1;
    }
    if (pos.size() >= 3 && atoi(pos[2]) != 0) {
        fprintf(stderr, "Unsupported Memory Type\n");
        return 1;
    }
    if (pos.size() >= 4) scene_path = pos[3];

    std::vector<rt_sphere> spheres(16384);
    uint32_t n = 0;
    rt_camera cam{};
    if (!scene_path.empty()) {
        if (rt_read_scene(scene_path.c_str(), spheres.data(), (uint32_t)spheres.size(), &n, &cam.orig,
                          &cam.target, doubling ? 1 : 0) != RT_OK)
            return die("readScene");
    } else {                                        // Main.cpp:80-86
        n = (uint32_t)rt_demo_scene(spheres.data(), (uint32_t)spheres.size());
        cam.orig = rt_vec3{ 20.f, 100.f, 120.f };
        cam.target = rt_vec3{ 0.f, 25.f, 0.f };
    }
    rt_compute_camera(&cam, w, h);

    if (oneshot > 0) {                               // the one call north_star names, as a host would use it per frame
        std::vector<uint32_t> px1(static_cast<size_t>(w) * h);
        const rt_scene scene{ spheres.data(), n };
        printf("{\"rt_render_wall_ms\": [");
        for (int k = 0; k < oneshot; ++k) {
            const auto t0 = std::chrono::steady_clock::now();
            if (rt_render(&scene, &cam, px1.data(), w, h, spp) != RT_OK) return die("rt_render");
            printf("%s%.3f", k ? ", " : "", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        printf("], \"w\": %d, \"h\": %d, \"spp\": %d, \"spheres\": %u}\n", w, h, spp, n);
        if (!out.empty() && !write_ppm(out, px1, w, h)) fprintf(stderr, "cannot write %s\n", out.c_str());
        rt_release_cache();
        return 0;
    }

    rt_ctx* ctx = nullptr;
    if (rehearse > 0) {
        std::vector<int> dev(static_cast<size_t>(rehearse), 0);
        if (rt_create_multi_on(&ctx, w, h, dev.data(), rehearse, 8) != RT_OK) return die("rt_create_multi_on");
    } else if ((gpus > 1 ? rt_create_multi(&ctx, w, h, gpus) : rt_create(&ctx, w, h)) != RT_OK) {
        return die("rt_create");
    }
    if (rt_set_scene(ctx, spheres.data(), n) != RT_OK) return die("rt_set_scene");
    if (rt_set_camera(ctx, &cam) != RT_OK) return die("rt_set_camera");
    if (rt_set_mode(ctx, mode) != RT_OK) return die("rt_set_mode");

    std::vector<uint32_t> px(static_cast<size_t>(w) * h);
    if (per_launch <= 0) per_launch = spp;
    if (pin && rt_pin_output(ctx, px.data(), px.size()) != RT_OK) return die("rt_pin_output");
    auto t0 = std::chrono::steady_clock::now();
    auto last_copy = t0;
    double kernel_ms = 0.0;
    for (int done = 0; done < spp;) {
        int k = (spp - done < per_launch) ? spp - done : per_launch;
        const auto now = std::chrono::steady_clock::now();
        const bool last = done + k >= spp;
        const bool due = done == 0 || last || readback_ms <= 0.0 ||
                         std::chrono::duration<double, std::milli>(now - last_copy).count() >= readback_ms;
        if (readback_ms > 0.0) rt_set_pixel_write(ctx, due ? 1 : 0);
        if ((due ? rt_render_pass(ctx, px.data(), k) : rt_render_async(ctx, k, rt_stream(ctx))) != RT_OK)
            return die("rt_render_pass");
        if (due) {                       // (launches queued between two copies are not timed one by one)
            last_copy = now;
            rt_stats st;
            rt_get_stats(ctx, &st);
            kernel_ms += st.last_kernel_ms;
        }
        done += k;
    }
    double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    rt_stats st;
    if (rt_get_stats(ctx, &st) != RT_OK) return die("rt_get_stats");
    if (!out.empty() && !write_ppm(out, px, w, h)) fprintf(stderr, "cannot write %s\n", out.c_str());

    const double rays = (double)(st.samples + st.shadow_rays);
    printf("{\"spheres\": %u, \"w\": %d, \"h\": %d, \"spp\": %d, \"launches\": %llu, \"kernel_ms\": %.4f, "
           "\"wall_ms_with_readback\": %.4f, \"samples\": %llu, \"closest_rays\": %llu, \"shadow_rays\": %llu, "
           "\"sphere_tests\": %llu, \"Mray_s_primary_shadow\": %.1f, \"Msample_s\": %.1f}\n",
           n, w, h, spp, (unsigned long long)st.launches, kernel_ms, wall_ms, (unsigned long long)st.samples,
           (unsigned long long)st.closest_rays, (unsigned long long)st.shadow_rays,
           (unsigned long long)st.sphere_tests, rays / (kernel_ms * 1e3), (double)st.samples / (kernel_ms * 1e3));
    rt_destroy(ctx);
    return 0;
}
