// HipConfig.cpp -- see HipConfig.hpp.  Mirrors OpenCLConfigBuffer (OpenCLConfig.cpp:398-747)
// call for call; every rt_* failure becomes the reference's fprintf(stderr) + exit(-1).
#include "HipConfig.hpp"

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

unsigned* HipConfig::getPixels() { return pPixels; }

void HipConfig::setArguments() {}                        // OpenCLConfig.cpp:517-611

void HipConfig::execute() {                              // OpenCLConfig.cpp:407-515
    if (rt_render_pass(ctx, reinterpret_cast<uint32_t*>(pPixels), 1) != RT_OK) die("Failed to render a pass");
}
