// ref_host_main.cpp -- builds oracle/_ref/ref_host_hip: the REFERENCE's own host code around the HIP
// backend.  TEST INFRASTRUCTURE (container build only; the binary is git-ignored and travels to the
// GPU box like oracle/_ref/libref.so).
//
// Linked from the reference's sources where they lie (/root/reference/SimpleRT/src/{Utility,Vec,Scene}.cpp
// unmodified; Config.cpp with adapter/reference_factory.patch applied to a scratch copy, oracle/Makefile) +
// adapter/HipConfig.cpp + librt_hip.so.  The flow below is the one of SimpleRT/src/Main.cpp:29-102 without the
// freeglut window: the backend from the reference's own factory -- createConfig(w, h, selectType(2), ...), framework
// ID 2 = the slot the patch fills (Config.cpp:13-68,99-110) --, the scene from `readScene` or `DemoSpheres`,
// `sceneSetup`, `updateCamera`, then N calls of `Config::updateRendering()` (the reference's pass driver,
// Config.cpp:73-91, incl. its caption), and the frame `getPixels()` returns, written as a PPM.
//
//   ref_host_hip <passes> <width> <height> <out.ppm> [scene.scn]
//   RT_TEST_MOVE_CAMERA_AT=K   see below
//   RT_TEST_CAPTION_LOG=file   every pass's caption ("... Sample/sec %.1fK") appended to `file`, and a last line
//                              "TRUE <samples per second over the whole loop, the drain included>"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "Config.hpp"
#include "Scene.hpp"
#include "Sphere.hpp"
#include "Utility.hpp"
#include "Vec.hpp"


int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <passes> <width> <height> <out.ppm> [scene.scn]\n", argv[0]);
        return 2;
    }
    const int passes = atoi(argv[1]), w = atoi(argv[2]), h = atoi(argv[3]);
    std::unique_ptr<Config> config = createConfig(w, h, selectType(2), true, MemType::Buffer);       // Main.cpp:29-66 with framework ID 2

    Vec orig, target;
    std::vector<Sphere> spheres;
    if (argc >= 6) {
        spheres = readScene(argv[5], orig, target);                      // Main.cpp:74-76
    } else {
        orig = { 20.f, 100.f, 120.f };                                    // Main.cpp:80-84
        target = { 0.f, 25.f, 0.f };
        spheres = DemoSpheres;
    }
    config->sceneSetup(spheres, orig, target);                            // Main.cpp:89
    config->updateCamera();                                               // Main.cpp:90

    char caption[256] = "";
    config->setCaptionBuffer(caption);                                    // SetupGL.cpp:83
    // RT_TEST_MOVE_CAMERA_AT=K (tests only): after K passes the eye moves by (5, 3, -4) and the accumulation goes on --
    // passes the backend had only counted or queued at that point belong to the OLD camera
    const char* move_at = getenv("RT_TEST_MOVE_CAMERA_AT");
    const int k_move = move_at ? atoi(move_at) : -1;
    const char* cap_path = getenv("RT_TEST_CAPTION_LOG");
    FILE* cap = cap_path ? fopen(cap_path, "w") : nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    for (int i = 0; i < passes; ++i) {                                    // Main.cpp:96-102
        if (i == k_move) {
            orig = { orig.x + 5.f, orig.y + 3.f, orig.z - 4.f };
            config->sceneSetup(spheres, orig, target);
            config->updateCamera();
        }
        config->updateRendering();
        if (cap) fputs(caption, cap);
    }
    fprintf(stderr, "%s", caption);

    const unsigned* px = config->getPixels();                             // SetupGL.cpp:85 (waits for what is queued)
    if (cap) {
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        fprintf(cap, "TRUE %.1f\n", (double)passes * w * h / sec);
        fclose(cap);
    }
    FILE* f = fopen(argv[4], "wb");
    if (!f) return 1;
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    for (int y = h - 1; y >= 0; --y)                                      // buffer row 0 is the bottom of the image
        for (int x = 0; x < w; ++x) {
            const unsigned p = px[(size_t)y * w + x];
            const unsigned char rgb[3] = { (unsigned char)(p & 255), (unsigned char)((p >> 8) & 255), (unsigned char)((p >> 16) & 255) };
            fwrite(rgb, 1, 3, f);
        }
    fclose(f);
    return 0;
}
