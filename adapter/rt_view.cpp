// rt_view.cpp -- the display component: the reference's window (SimpleRT/src/SetupGL.cpp:52-100,
// Main.cpp:18-113) around the HIP render path.  Same positional arguments as the reference's Main.cpp
//     rt_view <framework ID> <CPU/GPU> <mem> [scene.scn]      (framework ID 2 = this backend, Config.cpp:99-110)
// same window (800x600, SetupGL.cpp:32-33), same title and caption lines, glDrawPixels of the RGBX frame.
// Two differences, both on purpose:
//   * the frame the GL thread draws is a COMPLETE frame taken from a FrameExchange on every redraw, not one
//     pointer captured at start-up into a buffer the compute thread is writing (SetupGL.cpp:59-63,85);
//   * the caption carries the ray rate (Mray/s, rt_get_stats) beside the reference's Sample/sec (Config.cpp:84-88).
// Optional component: adapter/CMakeLists.txt builds it only where GLUT is found (this image has the GL and
// freeglut headers but no libglut, so here it is syntax-checked only: tests/test_adapter.py).
#ifdef __APPLE__
#include <GLUT/glut.h>
#else
#include <GL/glut.h>
#endif

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ProgressiveRenderer.hpp"
#include "rt_api.h"

static ProgressiveRenderer* renderer;
static const int kWinW = 800, kWinH = 600;               // the reference's window size (glWidth, glHeight: SetupGL.cpp:32-33)
static char caption_text[256];

static void draw_text(void* font, const char* text) {     // what SetupGL.cpp:42-47 (PrintString) does
    while (*text) glutBitmapCharacter(font, *text++);
}

static void on_idle(void) { glutPostRedisplay(); }        // idleFunc, SetupGL.cpp:52-57

static void on_display(void) {                           // displayFunc, SetupGL.cpp:59-76
    const uint32_t* frame = renderer->frames().acquire();    // the latest whole frame; never one being copied into
    renderer->copy_caption(caption_text, sizeof caption_text);
    glClear(GL_COLOR_BUFFER_BIT);
    glRasterPos2i(0, 0);
    glDrawPixels(kWinW, kWinH, GL_RGBA, GL_UNSIGNED_BYTE, frame);   // row 0 = bottom of the image, byte 0 = R (.cl:594-596)
    glColor3f(1.f, 1.f, 1.f);
    glRasterPos2i(4, kWinH - 16);
    draw_text(GLUT_BITMAP_HELVETICA_18, "Ray Tracing Experiment");
    glColor3f(1.f, 1.f, 1.f);
    glRasterPos2i(4, 10);
    draw_text(GLUT_BITMAP_HELVETICA_18, caption_text);
    glutSwapBuffers();
}

int main(int argc, char* argv[]) {
    fprintf(stderr, "Usage: %s <framework ID> <CPU/GPU> <mem> <scene file>   (framework ID 2 = HIP)\n", argv[0]);   // Main.cpp:24-26
    if (argc > 1 && atoi(argv[1]) != 2) {
        fprintf(stderr, "Unsupported Framework Type\n");      // Config.cpp:63-65
        return -1;
    }
    if (argc > 3 && atoi(argv[3]) != 0) {
        fprintf(stderr, "Unsupported Memory Type\n");
        return -1;
    }
    std::vector<rt_sphere> spheres(2 * RT_MAX_SPHERES);
    uint32_t count = 0;
    rt_camera cam{};
    if (argc > 4) {                                           // Main.cpp:75-79: readScene, with the loader's 2N doubling
        if (rt_read_scene(argv[4], spheres.data(), static_cast<uint32_t>(spheres.size()), &count, &cam.orig, &cam.target, 1) != RT_OK) {
            fprintf(stderr, "Failed to read scene: %s\n", rt_last_error());
            return -1;
        }
    } else {                                                  // Main.cpp:80-86
        count = static_cast<uint32_t>(rt_demo_scene(spheres.data(), static_cast<uint32_t>(spheres.size())));
        cam.orig = rt_vec3{ 20.f, 100.f, 120.f };
        cam.target = rt_vec3{ 0.f, 25.f, 0.f };
    }
    rt_compute_camera(&cam, kWinW, kWinH);               // updateCamera, OpenCLConfig.cpp:386-392

    rt_ctx* ctx = nullptr;
    if (rt_create(&ctx, kWinW, kWinH) != RT_OK || rt_set_scene(ctx, spheres.data(), count) != RT_OK ||
        rt_set_camera(ctx, &cam) != RT_OK) {
        fprintf(stderr, "Failed to set up the HIP render context: %s\n", rt_last_error());
        exit(-1);
    }
    const char* ms = getenv("RT_READBACK_MS");
    ProgressiveRenderer progressive(ctx, kWinW, kWinH, ms ? atof(ms) : 8.0, 0);
    renderer = &progressive;
    progressive.start();                                      // the compute thread, Main.cpp:96-102

    char title[] = "SimpleRT (HIP / MI355X)";                 // InitGlut, SetupGL.cpp:80-100
    glutInitWindowSize(kWinW, kWinH);
    glutInitWindowPosition(0, 0);
    glutInitDisplayMode(GLUT_RGB | GLUT_DOUBLE);
    glutInit(&argc, argv);
    glutCreateWindow(title);
    glutDisplayFunc(on_display);
    glutIdleFunc(on_idle);
    glViewport(0, 0, kWinW, kWinH);
    glLoadIdentity();
    glOrtho(0.f, kWinW - 1.f, 0.f, kWinH - 1.f, -1.f, 1.f);
    glutMainLoop();                                           // Main.cpp:106 (does not return)
    progressive.stop();
    rt_destroy(ctx);
    return 0;
}
