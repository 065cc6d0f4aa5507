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

void HipConfig::sceneSetup(const std::vector<Sphere>& spheres, Vec orig, Vec target) {   // :720-747
    if (rt_set_scene(ctx, reinterpret_cast<const rt_sphere*>(spheres.data()),
                     static_cast<uint32_t>(spheres.size())) != RT_OK)
        die("Failed to upload the scene");
    camera.orig = orig;
    camera.target = target;
}

void HipConfig::updateCamera() {                         // OpenCLConfig.cpp:386-392
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
        if (rt_read_pixels(ctx, reinterpret_cast<uint32_t*>(pPixels)) != RT_OK) die("Failed to read the frame back");
        stale = false;
    }
    return pPixels;
}

void HipConfig::setArguments() {}                        // OpenCLConfig.cpp:517-611

// One pass.  The reference reads the frame back after EVERY pass (OpenCLConfig.cpp:498-512) although
// the window only looks at it when it redraws (SetupGL.cpp:59-63, unsynchronised).  A pass takes
// 0.05-0.12 ms on an MI355X and the 8.3 MB readback of a 1080p frame three times that, so the frame
// is copied when it is due for display: on pass 0 and then every readbackMs (8 ms unless
// RT_READBACK_MS says otherwise; 0 = after every pass, the reference's cadence).  The passes in
// between are only queued (rt_render_async on the context's stream); the copy that is due waits for
// them, which also bounds the queue to readbackMs of work.
void HipConfig::execute() {                              // OpenCLConfig.cpp:407-515
    std::lock_guard<std::mutex> lock(guard);
    const auto now = std::chrono::steady_clock::now();
    const bool due = mCurrentSample == 0 || readbackMs <= 0.0 ||
                     std::chrono::duration<double, std::milli>(now - lastReadback).count() >= readbackMs;
    rt_set_pixel_write(ctx, due ? 1 : 0);       // passes nobody looks at skip the gamma + pixel store
    const int rc = due ? rt_render_pass(ctx, reinterpret_cast<uint32_t*>(pPixels), 1)
                       : rt_render_async(ctx, 1, rt_stream(ctx));
    if (rc != RT_OK) die("Failed to render a pass");
    stale = !due;
    if (due) lastReadback = now;
}
