// HipConfig.cpp -- see HipConfig.hpp.  Mirrors OpenCLConfigBuffer (OpenCLConfig.cpp:398-747)
// call for call; every rt_* failure becomes the reference's fprintf(stderr) + exit(-1).
#include "HipConfig.hpp"

#include <chrono>
#include <thread>
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
    if (rt_throttle(ctx, 2, nullptr) != RT_OK) die("Failed to set up pacing");      // (the first call switches the bookkeeping on)
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
        copyQueued = false;
    } else if (copyQueued) {
        if (rt_throttle(ctx, 0, nullptr) != RT_OK) die("Failed to wait for the frame copy");     // the display copy in flight lands first
        copyQueued = false;
    }
    return pPixels;
}

void HipConfig::setArguments() {}                        // OpenCLConfig.cpp:517-611

// One pass.  The reference reads the frame back after EVERY pass (OpenCLConfig.cpp:498-512) although the window only
// looks at it when it redraws (SetupGL.cpp:59-63, unsynchronised).  A pass takes 0.04-0.12 ms on an MI355X, the 8.3 MB
// readback of a 1080p frame three times that, and a launch of one pass costs twice what the pass costs inside a longer
// launch.  So passes are counted here and launched `batch` at a time (about a millisecond of work per launch, without
// gamma and pixel store), and the frame is copied when it is due for display -- on pass 0 and then every readbackMs
// (8 ms unless RT_READBACK_MS says otherwise; 0 = the reference's cadence: one blocking pass + copy per call).
//
// The caller times this function: Config::updateRendering (Config.cpp:73-91, not virtual) puts W*H / elapsed into the
// window caption as "Sample/sec".  For that figure to stay TRUE with batching, a call must last what a pass costs the
// device, not the microsecond it takes to count it:
//   * the queue is bounded -- after a launch the call waits until at most two launches are unfinished (rt_throttle),
//     which also returns the device time per pass of the launch that finished last;
//   * every call then lasts at least 0.9 x that time (this is the compute thread, which the reference blocks in clFinish: a
//     spin of tens of microseconds for the usual pass, a sleep with a 150 us spin at its end for passes beyond 0.3 ms), so
//     the host runs at the device's pace, a fraction ahead; the launching call takes up the slack.
// The frame that is due is copied asynchronously into pPixels (page-locked) behind the passes -- the window reads that
// buffer unsynchronised in the reference too -- so a due pass costs a launch, not a drain of the queue.
void HipConfig::execute() {                              // OpenCLConfig.cpp:407-515
    std::lock_guard<std::mutex> lock(guard);
    const auto t0 = std::chrono::steady_clock::now();
    if (readbackMs <= 0.0) {                             // the reference's own cadence
        if (rt_set_pixel_write(ctx, 1) != RT_OK || rt_render_pass(ctx, reinterpret_cast<uint32_t*>(pPixels), 1) != RT_OK)
            die("Failed to render a pass");
        stale = false;
        return;
    }
    const bool due = mCurrentSample == 0 || std::chrono::duration<double, std::milli>(t0 - lastReadback).count() >= readbackMs;
    pending += 1;
    stale = true;
    if (due || pending >= batch) {
        launchPending(due);
        if (due) {
            if (rt_read_pixels_async(ctx, reinterpret_cast<uint32_t*>(pPixels), rt_stream(ctx)) != RT_OK) die("Failed to queue the frame copy");
            copyQueued = true;
            stale = false;                              // the copy in flight is the frame of every pass so far
            lastReadback = t0;
        }
        double ms = 0.0;
        if (rt_throttle(ctx, 2, &ms) != RT_OK) die("Failed to wait for the device");
        if (ms > 0.0) {
            passMs = ms;
            const int want = static_cast<int>(1.0 / ms + 0.5);          // about a millisecond of work per launch
            batch = want < 1 ? 1 : (want > 64 ? 64 : want);
        }
    }
    if (passMs > 0.0) {
        const auto until = t0 + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double, std::milli>(0.9 * passMs));
        // Passes of tens of microseconds are spun out; a pass that costs the device a millisecond does not hold a core for it: the thread
        // sleeps through all but the last 150 us (a wake-up comes some tens of microseconds late) and spins only those.
        const auto left = until - std::chrono::steady_clock::now();
        if (left > std::chrono::microseconds(300)) std::this_thread::sleep_for(left - std::chrono::microseconds(150));
        while (std::chrono::steady_clock::now() < until) { /* spin: tens of microseconds */ }
    }
}
