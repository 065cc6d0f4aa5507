// view_headless.cpp -- the display component without a window: ProgressiveRenderer (the compute thread of
// adapter/rt_view.cpp) with a reader thread in the place of displayFunc.  The reader takes frames from the
// FrameExchange as fast as it can while passes are being rendered and copied, and prints one line per
// DISTINCT frame it saw: the pass count the frame was published with and an FNV-1a hash of its pixels.
// tests/test_adapter.py compares each with the oracle's frame of that pass count: a torn frame, a frame
// still being copied into, or a frame of the wrong pass would not match.
//   view_headless <w> <h> <passes> <readback_ms>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../adapter/ProgressiveRenderer.hpp"
#include "rt_api.h"

static unsigned long long fnv(const uint32_t* p, size_t n) {
    unsigned long long hsh = 0xcbf29ce484222325ull;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(p);
    for (size_t i = 0; i < 4 * n; ++i) hsh = (hsh ^ b[i]) * 0x100000001b3ull;
    return hsh;
}

int main(int argc, char** argv) {
    const int w = argc > 1 ? atoi(argv[1]) : 160, h = argc > 2 ? atoi(argv[2]) : 96, passes = argc > 3 ? atoi(argv[3]) : 200;
    const double readback_ms = argc > 4 ? atof(argv[4]) : 0.0;
    std::vector<rt_sphere> sph(16);
    const uint32_t n = static_cast<uint32_t>(rt_demo_scene(sph.data(), 16));
    rt_camera cam{};
    cam.orig = rt_vec3{ 20.f, 100.f, 120.f };
    cam.target = rt_vec3{ 0.f, 25.f, 0.f };
    rt_compute_camera(&cam, w, h);
    rt_ctx* ctx = nullptr;
    if (rt_create(&ctx, w, h) != RT_OK || rt_set_scene(ctx, sph.data(), n) != RT_OK || rt_set_camera(ctx, &cam) != RT_OK) {
        fprintf(stderr, "setup failed: %s\n", rt_last_error());
        return 1;
    }
    ProgressiveRenderer progressive(ctx, w, h, readback_ms, passes);
    unsigned long long seen = 0, reads = 0;
    uint64_t last_seq = 0;
    progressive.start();
    for (;;) {
        const bool finished = progressive.finished();         // read BEFORE acquiring: the final frame is then never missed
        uint64_t seq = 0;
        bool fresh = false;
        const uint32_t* frame = progressive.frames().acquire(&seq, &fresh);
        ++reads;
        if (fresh && seq != 0) {
            if (seq <= last_seq) {
                fprintf(stderr, "frame sequence went backwards: %llu after %llu\n", (unsigned long long)seq, (unsigned long long)last_seq);
                return 2;
            }
            last_seq = seq;
            printf("{\"pass\": %llu, \"fnv\": \"%016llx\"}\n", (unsigned long long)seq, fnv(frame, static_cast<size_t>(w) * h));
            ++seen;
        }
        if (finished) break;
    }
    progressive.stop();
    char caption[256];
    progressive.copy_caption(caption, sizeof caption);
    for (char* c = caption; *c; ++c)
        if (*c == '\n' || *c == '"') *c = ' ';
    printf("{\"frames_seen\": %llu, \"reads\": %llu, \"last_pass\": %llu, \"failed\": %d, \"caption\": \"%s\"}\n", seen, reads,
           (unsigned long long)last_seq, progressive.failed() ? 1 : 0, caption);
    rt_destroy(ctx);
    return progressive.failed() ? 3 : 0;
}
