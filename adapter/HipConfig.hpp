// HipConfig.hpp -- the reference-side binding: a `Config` backend (SimpleRT/include/Config.hpp:11-40)
// that renders through the C ABI of include/rt_api.h instead of OpenCL.  It occupies the slot the
// reference leaves unimplemented: SupportType::Default, framework ID 2 (Config.cpp:63-65,99-110).
//
// Compiled only inside a checkout of the reference (it includes the reference's own headers);
// INTEGRATION.md lists the three edits the reference needs (factory case, CMake, link line).
#ifndef HIP_CONFIG_HPP
#define HIP_CONFIG_HPP

#include <chrono>
#include <mutex>
#include <memory>      // the factory (Config.cpp) uses std::make_unique / std::runtime_error and gets
#include <stdexcept>   // them from the OpenCL / Cm headers today; with only this backend enabled, from here
#include <vector>

#include "Camera.hpp"   // reference headers
#include "Config.hpp"
#include "Sphere.hpp"
#include "Vec.hpp"

#include "rt_api.h"

class HipConfig : public Config {
public:
    HipConfig(int width, int height);
    ~HipConfig() override;

    void sceneSetup(const std::vector<Sphere>& spheres, Vec orig, Vec target) override;
    void updateCamera() override;
    unsigned* getPixels() override;

private:
    void setArguments() override;   // nothing to bind: the context holds the arguments
    void execute() override;        // one pass (counted; launched in batches between two display copies)
    void allocateBuffer() override;
    void freeBuffer() override;

    [[noreturn]] void die(const char* what) const;   // the reference's print-and-exit policy

    rt_ctx* ctx = nullptr;
    unsigned* pPixels = nullptr;    // stable for the life of the object (SetupGL.cpp:85 keeps it)
    Camera camera{};
    double readbackMs = 8.0;        // copy the frame to pPixels when it is this old (0 = every pass)
    std::chrono::steady_clock::time_point lastReadback{};
    bool stale = false;             // passes have run since pPixels was last brought up to date
    int pending = 0;                // passes execute() has counted and not launched yet
    int batch = 16;                 // launch them this many at a time (about a millisecond of work: set from the measured pass time)
    double passMs = 0.0;            // device time per pass of the launch that finished last (0 = not known yet): what a call lasts
    bool copyQueued = false;        // an asynchronous display copy into pPixels may still be in flight
    void launchPending(bool withPixels);
    std::mutex guard;               // execute() runs on the compute thread, getPixels() on the caller's (Main.cpp:96-106)
};

#endif
