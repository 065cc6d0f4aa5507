// ProgressiveRenderer.hpp -- the reference's compute thread (Main.cpp:96-102: `while (true)
// updateRendering();`) on the C ABI, publishing whole frames through a FrameExchange and keeping the
// caption of Config::updateRendering (Config.cpp:84-88) extended with the ray rate from rt_get_stats.
// Shared by the GLUT viewer (adapter/rt_view.cpp) and its headless twin (tools/view_headless.cpp).
#ifndef PROGRESSIVE_RENDERER_HPP
#define PROGRESSIVE_RENDERER_HPP

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

#include "FrameExchange.hpp"
#include "rt_api.h"

class ProgressiveRenderer {
public:
    ProgressiveRenderer(rt_ctx* context, int width, int height, double readback_ms_, int max_passes_)
        : ctx(context), w(width), h(height), readback_ms(readback_ms_), max_passes(max_passes_),
          exchange(static_cast<size_t>(width) * height) {
        // one registration for the three frames: the per-frame read-back then runs at the PCIe rate (optional)
        (void)rt_pin_output(ctx, exchange.storage(), exchange.storage_count());
        snprintf(caption, sizeof caption, "Rendering...");
    }
    ~ProgressiveRenderer() { stop(); }

    void start() { worker = std::thread([this] { loop(); }); }
    void stop() {
        quit.store(true);
        if (worker.joinable()) worker.join();
    }
    bool finished() const { return done.load(); }
    bool failed() const { return error.load(); }
    FrameExchange& frames() { return exchange; }
    void copy_caption(char* dst, size_t cap) {               // display thread
        std::lock_guard<std::mutex> lock(caption_mu);
        snprintf(dst, cap, "%s", caption);
    }

private:
    using clock = std::chrono::steady_clock;

    // Config::updateRendering's loop.  A pass takes 0.04-0.12 ms on an MI355X and the read-back of a 1080p frame
    // several times that, so a frame is copied out when it is due for display (pass 0, then every readback_ms;
    // 0 = after every pass, the reference's cadence); the passes in between are queued `batch` per launch (about a
    // millisecond of work: a launch of one pass costs twice what the pass costs inside a longer launch), without
    // the gamma/pack step nobody would look at (rt_set_pixel_write(0)).
    void loop() {
        auto last_copy = clock::now(), caption_t0 = last_copy;
        rt_stats prev{};
        int pass = 0, caption_pass0 = 0, batch = 1;
        while (!quit.load() && (max_passes <= 0 || pass < max_passes)) {
            const auto now = clock::now();
            const int left = max_passes > 0 ? max_passes - pass : batch + 1;
            const bool due = pass == 0 || left == 1 || readback_ms <= 0.0 ||
                             std::chrono::duration<double, std::milli>(now - last_copy).count() >= readback_ms;
            const int n = due ? 1 : (batch < left - 1 ? batch : left - 1);       // (the last pass is a due one)
            int rc = rt_set_pixel_write(ctx, due ? 1 : 0);
            if (rc == RT_OK) rc = due ? rt_render_pass(ctx, exchange.back(), 1) : rt_render_async(ctx, n, rt_stream(ctx));
            if (rc != RT_OK) {
                fprintf(stderr, "Failed to render a pass: %s\n", rt_last_error());
                error.store(true);
                break;
            }
            pass += n;
            if (due) {
                exchange.publish(static_cast<uint64_t>(pass));
                last_copy = now;
                rt_stats st{};
                if (rt_get_stats(ctx, &st) == RT_OK) {       // everything queued has completed: rt_render_pass waited
                    const double sec = std::chrono::duration<double>(clock::now() - caption_t0).count();
                    const double samples = static_cast<double>(st.samples - prev.samples);
                    const double rays = samples + static_cast<double>(st.shadow_rays - prev.shadow_rays);
                    std::lock_guard<std::mutex> lock(caption_mu);
                    snprintf(caption, sizeof caption, "Rendering time %.3f sec (pass %d)  Sample/sec  %.1fK  %.1f Mray/s\n",
                             sec / (pass - caption_pass0 > 0 ? pass - caption_pass0 : 1), pass, samples / sec / 1000.0, rays / sec / 1e6);
                    prev = st;
                    caption_t0 = clock::now();
                    caption_pass0 = pass;
                    if (st.last_kernel_ms > 0.0 && readback_ms > 0.0) {      // the one-pass launch just timed: size the batches
                        const int want = static_cast<int>(1.0 / st.last_kernel_ms + 0.5);
                        batch = want < 1 ? 1 : (want > 64 ? 64 : want);
                    }
                }
            }
        }
        done.store(true);
    }

    rt_ctx* ctx;
    int w, h;
    double readback_ms;
    int max_passes;
    FrameExchange exchange;
    std::thread worker;
    std::atomic<bool> quit{ false }, done{ false }, error{ false };
    std::mutex caption_mu;
    char caption[256];
};

#endif
