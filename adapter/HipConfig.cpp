// HipConfig.cpp -- see HipConfig.hpp.  Mirrors OpenCLConfigBuffer (OpenCLConfig.cpp:398-747)
// call for call; every rt_* failure becomes the reference's fprintf(stderr) + exit(-1).
#include "HipConfig.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "Utility.hpp"   // computeCameraVariables

static_assert(sizeof(Sphere) == sizeof(rt_sphere), "Sphere layout is the ABI (44 bytes)");
static_assert(sizeof(Camera) == sizeof(rt_camera), "Camera layout is the ABI (60 bytes)");

HipConfig::HipConfig(int width, int height) : Config(width, height) { allocateBuffer(); }

HipConfig::~HipConfig() { freeBuffer(); }

void HipConfig::die(const char* what) const {
    fprintf(stderr, "%s: %s\n", what, rt_last_error());
    exit(-1);
}

void HipConfig::allocateBuffer() {                       // OpenCLConfig.cpp:613-682
    pPixels = new unsigned[static_cast<size_t>(mWidth) * mHeight]();
    if (rt_create(&ctx, mWidth, mHeight) != RT_OK) die("Failed to create the HIP render context");
    // the per-pass readback into pPixels then runs at the PCIe rate (optional; failure is not fatal)
    (void)rt_pin_output(ctx, reinterpret_cast<uint32_t*>(pPixels), static_cast<size_t>(mWidth) * mHeight);
    if (const char* v = getenv("RT_READBACK_MS")) readbackMs = atof(v);
}

void HipConfig::freeBuffer() {                           // OpenCLConfig.cpp:684-717
    rt_destroy(ctx);
    ctx = nullptr;
    delete[] pPixels;
    pPixels = nullptr;
}

// Passes that execute() has counted but not launched belong to the scene and camera they were asked under.
void HipConfig::launchPending(bool withPixels) {
    if (pending == 0) return;
    if (rt_set_pixel_write(ctx, withPixels ? 1 : 0) != RT_OK ||
        rt_render_async(ctx, pending, rt_stream(ctx)) != RT_OK)
        die("Failed to render a pass");
    pending = 0;
}

void HipConfig::sceneSetup(const std::vector<Sphere>& spheres, Vec orig, Vec target) {   // :720-747
    std::lock_guard<std::mutex> lock(guard);
    launchPending(false);
    if (rt_set_scene(ctx, reinterpret_cast<const rt_sphere*>(spheres.data()),
                     static_cast<uint32_t>(spheres.size())) != RT_OK)
        die("Failed to upload the scene");
    camera.orig = orig;
    camera.target = target;
}

void HipConfig::updateCamera() {                         // OpenCLConfig.cpp:386-392
    std::lock_guard<std::mutex> lock(guard);
    launchPending(false);
    computeCameraVariables(&camera, mWidth, mHeight);
    rt_camera c;
    memcpy(&c, &camera, sizeof c);
    if (rt_set_camera(ctx, &c) != RT_OK) die("Failed to set the camera");
}

// The reference copies the frame out after every pass, so its getPixels() always names the latest frame
// (OpenCLConfig.cpp:498-512).  Here passes between two display copies are only queued: a caller that asks
// for the pixels gets them brought up to date first -- rt_read_pixels waits for the queued passes, packs the
// frame from the running average if their pixel stores were skipped, and copies it.  (The GLUT flow asks
// once, before the first frame, SetupGL.cpp:85; hosts that render N passes and then read, like
// oracle/ref_host_main.cpp, get the N-pass frame.)  The address stays the same for the life of the object.
unsigned* HipConfig::getPixels() {
    std::lock_guard<std::mutex> lock(guard);
    if (stale) {
        launchPending(true);
        if (rt_read_pixels(ctx, reinterpret_cast<uint32_t*>(pPixels)) != RT_OK) die("Failed to read the frame back");
        stale = false;
    }
    return pPixels;
}

void HipConfig::setArguments() {}                        // OpenCLConfig.cpp:517-611

// One pass.  The reference reads the frame back after EVERY pass (OpenCLConfig.cpp:498-512) although
// the window only looks at it when it redraws (SetupGL.cpp:59-63, unsynchronised).  A pass takes
// 0.04-0.12 ms on an MI355X, the 8.3 MB readback of a 1080p frame three times that, and a launch of one pass
// costs twice what the pass costs inside a longer launch (prologue, epilogue, the seed / colour round trip through
// HBM).  So: the frame is copied when it is due for display -- on pass 0 and then every readbackMs (8 ms unless
// RT_READBACK_MS says otherwise; 0 = after every pass, the reference's cadence) -- and the passes in between are
// counted here and launched `batch` at a time, about a millisecond of work per launch (1080p: 11 900 passes/s one
// per launch, 21 800 at 16).  The copy that is due launches what is pending and waits for everything queued, which
// also bounds the queue to readbackMs of work.  Scene and camera changes and getPixels() launch what is pending first.
void HipConfig::execute() {                              // OpenCLConfig.cpp:407-515
    std::lock_guard<std::mutex> lock(guard);
    const auto now = std::chrono::steady_clock::now();
    const bool due = mCurrentSample == 0 || readbackMs <= 0.0 ||
                     std::chrono::duration<double, std::milli>(now - lastReadback).count() >= readbackMs;
    pending += 1;
    if (due) {
        const int n = pending;
        pending = 0;
        if (rt_set_pixel_write(ctx, 1) != RT_OK ||
            rt_render_pass(ctx, reinterpret_cast<uint32_t*>(pPixels), n) != RT_OK)
            die("Failed to render a pass");
        stale = false;
        lastReadback = now;
        rt_stats st;                                    // size the batches from the launch that just finished
        if (n >= 4 && rt_get_stats(ctx, &st) == RT_OK && st.last_kernel_ms > 0.0) {
            const double per_pass = st.last_kernel_ms / n;
            const int want = static_cast<int>(1.0 / per_pass + 0.5);
            batch = want < 1 ? 1 : (want > 64 ? 64 : want);
        }
    } else {
        stale = true;
        if (pending >= batch) launchPending(false);     // passes nobody looks at skip the gamma + pixel store
    }
}
