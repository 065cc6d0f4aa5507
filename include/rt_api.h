/*
 * rt_api.h -- C ABI of the MI355X (gfx950) render path for RayTracing_Simple.
 *
 * This is the drop-in boundary: everything the reference's OpenCL backend does between
 * `Config::updateRendering()` and the pixel buffer -- the `RayTracing` kernel
 * (SimpleRT/kernel/RayTracing_Kernel.cl:551-600) and its launch wrapper
 * `OpenCLConfigBuffer` (SimpleRT/src/OpenCLConfig.cpp:398-747) -- behind plain C entry
 * points: POD structs, raw pointers and sizes, int status codes, no C++/torch types.
 * INTEGRATION.md shows the `HipConfig : Config` adapter the reference host would add.
 *
 * All file:line citations are relative to the reference checkout (SimpleRT/...).
 * Threading: calls on one rt_ctx are not re-entrant; any thread may make them (each entry
 * point selects the context's device itself).  Nothing here throws or exits.
 */
#ifndef RT_API_H
#define RT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- data layouts: identical, byte for byte, to the reference's host structs ---------- */
typedef struct { float x, y, z; } rt_vec3;                /* include/Vec.hpp:10-34   (12 B) */

enum { RT_DIFF = 0, RT_SPEC = 1, RT_REFR = 2 };           /* include/Sphere.hpp:6-8         */

typedef struct {                                          /* include/Sphere.hpp:11-15 (44 B) */
    float   rad;
    rt_vec3 p, e, c;                                      /* centre, emission, colour        */
    int32_t refl;                                         /* RT_DIFF / RT_SPEC / RT_REFR     */
} rt_sphere;

typedef struct {                                          /* include/Camera.hpp:7-14  (60 B) */
    rt_vec3 orig, target;                                 /* set by the user                 */
    rt_vec3 dir, x, y;                                    /* computeCameraVariables' output  */
} rt_camera;

typedef struct {                                          /* `sphere`, `sphereCount` kernel  */
    const rt_sphere *spheres;                             /*  arguments (.cl:553-554)        */
    uint32_t         count;
} rt_scene;

/* Exact work counters of everything rendered since rt_create()/rt_reset(). */
typedef struct {
    uint64_t samples;        /* camera (primary) rays                                         */
    uint64_t closest_rays;   /* closest-hit queries, `Intersect` .cl:215 (primary+extension)  */
    uint64_t shadow_rays;    /* any-hit queries, `IntersectP` .cl:234                         */
    uint64_t sphere_tests;   /* `SphereIntersect` .cl:173 evaluations                         */
    uint64_t rng_draws;      /* `GetRandom` .cl:143 calls                                     */
    uint64_t launches;       /* kernel launches                                               */
    double   last_kernel_ms; /* device time of the last rt_render_pass launch (HIP events)    */
} rt_stats;

enum rt_status {
    RT_OK            =  0,
    RT_ERR_ARG       = -1,   /* null pointer, non-positive size, bad enum, scene too large    */
    RT_ERR_NO_DEVICE = -2,   /* no usable gfx950 device / HIP runtime                         */
    RT_ERR_HIP       = -3,   /* a HIP call failed; rt_last_error() has the text               */
    RT_ERR_ALLOC     = -4,
    RT_ERR_STATE     = -5    /* render requested before a scene and a camera were set         */
};

enum rt_mode {
    RT_MODE_PARITY = 0,      /* strict binary32, no contraction, restated libm: bit-exact     */
    RT_MODE_FAST   = 1       /* FMA contraction + hardware rcp/rsq/sin/cos/exp2/log2           */
};

#define RT_MAX_SPHERES 8192u /* geometry table (16 B per sphere) must fit the CU's LDS        */

typedef struct rt_ctx rt_ctx;

/* ---- the headline call ------------------------------------------------------------------
 * Equivalent to: a fresh OpenCLConfigBuffer(w,h) [seeds = never-seeded std::rand() stream
 * clamped to >= 2, OpenCLConfig.cpp:676-680] + sceneSetup + `spp` calls of
 * Config::updateRendering() [Config.cpp:73-81], after which `out` holds what getPixels()
 * returns: out[y*w+x] = R | G<<8 | B<<16, row 0 = bottom of the image (.cl:594-596).
 * `cam` carries dir/x/y already computed by the host (Utility.cpp:71-85).
 * `out` is a host buffer of w*h uint32.  Parity mode, device 0, blocking.                    */
int rt_render(const rt_scene *scene, const rt_camera *cam, uint32_t *out, int w, int h, int spp);

/* ---- progressive interface: one context = one OpenCLConfigBuffer ------------------------ */

/* ctor + allocateBuffer (OpenCLConfig.cpp:398-400, 613-682): device buffers for colours
 * (12 B/px), seeds (8 B/px, initialised to the default stream) and pixels (4 B/px).         */
int rt_create(rt_ctx **out, int w, int h);

/* Same, on HIP device `device`, rendering only the row tiles this rank owns: tile t (rows
 * [t*tile_rows, (t+1)*tile_rows)) belongs to rank t % nranks.  tile_rows must be a positive
 * multiple of 8.  The rank's rows are packed in order into a local pixel buffer of
 * rt_local_rows() rows (interleaved row tiles).  nranks = 1 is rt_create().       */
int rt_create_sharded(rt_ctx **out, int w, int h, int device, int rank, int nranks,
                      int tile_rows);

void rt_destroy(rt_ctx *ctx);                              /* freeBuffer, :684-717            */

/* sceneSetup (:720-747) + the per-pass sphere upload (:450).  Copies; caller keeps `spheres`. */
int rt_set_scene(rt_ctx *ctx, const rt_sphere *spheres, uint32_t count);

/* updateCamera's result + the per-pass camera upload (:418).  dir/x/y must be filled in.     */
int rt_set_camera(rt_ctx *ctx, const rt_camera *cam);

int rt_set_mode(rt_ctx *ctx, int mode);                    /* enum rt_mode; default parity    */

/* Back to pass 0: mCurrentSample = 0, seeds = default stream, counters cleared.              */
int rt_reset(rt_ctx *ctx);

/* rt_reset() without a host round trip: the next launch starts from a device-resident copy of
 * the default stream (read in place, nothing is copied) and the counters are cleared by a small
 * kernel on `hip_stream`.  (The colour plane needs no clearing: pass 0 overwrites it,
 * .cl:580-582.)                                                                              */
int rt_reset_async(rt_ctx *ctx, void *hip_stream);

/* `n_samples` x { setArguments(); execute(); ++mCurrentSample; } (Config.cpp:73-81) as ONE
 * launch that keeps seeds and the running average in registers, then one D2H copy of the
 * pixel buffer into `out_host` (the full image for an unsharded context, the local rows for
 * a sharded one).  out_host may be NULL to skip the copy.  Blocking.                         */
int rt_render_pass(rt_ctx *ctx, uint32_t *out_host, int n_samples);

/* Page-lock the host buffer that rt_render_pass copies into (the host's `pPixels`,
 * OpenCLConfig.cpp:618-621), so the per-pass readback of the reference's display loop
 * (clEnqueueReadBuffer after every pass, OpenCLConfig.cpp:498-512) runs at the full PCIe rate
 * instead of through a pageable staging copy.  `count` uint32 from `out_host` must stay valid
 * and at the same address until rt_pin_output(ctx, NULL, 0) or rt_destroy.  Optional.        */
int rt_pin_output(rt_ctx *ctx, uint32_t *out_host, size_t count);

/* enable = 0: later launches advance seeds and the running average but leave the packed pixel
 * buffer alone (no toInt, .cl:34,594-596, and no pixel store) -- for passes whose frame nobody
 * will look at; the next launch with enable = 1 writes every pixel of its frame from the running
 * average, so nothing is lost.  Default 1.                                                    */
int rt_set_pixel_write(rt_ctx *ctx, int enable);

/* Same launch, asynchronous on `hip_stream` (a hipStream_t, NULL = default stream), no copy
 * and no synchronisation: the caller orders later work on that stream.                       */
int rt_render_async(rt_ctx *ctx, int n_samples, void *hip_stream);

/* The context's own non-blocking stream (a hipStream_t), the one rt_render_pass uses.  With
 * several contexts in flight, rt_render_async(ctx, n, rt_stream(ctx)) runs each on its own.
 * HIP gives a process GPU_MAX_HW_QUEUES hardware queues (default 4) and lets further streams
 * share them: two contexts on one queue do not overlap at all, so a host that keeps F contexts
 * in flight should start with GPU_MAX_HW_QUEUES >= the number of streams it uses (bench.py: 24). */
void *rt_stream(rt_ctx *ctx);

/* Device address and element count (uint32) of the local pixel buffer.                       */
int rt_device_pixels(rt_ctx *ctx, void **dptr, size_t *count);

/* Redirect the packed pixels of later launches into a caller-owned DEVICE buffer of at least
 * rt_local_rows()*w uint32 (e.g. the send buffer of the frame-end gather, so no copy is needed);
 * NULL restores the context's own buffer.  The caller keeps the buffer alive and orders its
 * reuse against the launches it issued.                                                      */
int rt_set_pixel_buffer(rt_ctx *ctx, void *dptr, size_t count);

int rt_local_rows(const rt_ctx *ctx);                      /* rows this context renders       */
int rt_current_sample(const rt_ctx *ctx);                  /* mCurrentSample                  */

/* Copies of the reference's other two buffers, for parity checks: the colour plane
 * (3 floats/px, y-flipped as .cl:579 stores it) and the seed pairs, full image size; rows
 * this rank does not own keep their initial content.                                         */
int rt_read_colors(rt_ctx *ctx, float *out_host);
int rt_read_seeds(rt_ctx *ctx, uint32_t *out_host);

int rt_get_stats(rt_ctx *ctx, rt_stats *out);

/* Text of the calling thread's last failure ("" if none).                                    */
const char *rt_last_error(void);

/* ---- host-side helpers either side of the path ---------------------------- */

/* computeCameraVariables, Utility.cpp:71-85 (Vec::norm's double sqrt, Vec.cpp:28-30).        */
void rt_compute_camera(rt_camera *cam, int w, int h);

/* The seed initialisation of OpenCLConfig.cpp:676-680 without depending on the host libc:
 * glibc's never-seeded rand() stream restated, each value clamped to >= 2.                   */
void rt_default_seeds(uint32_t *seeds, size_t count);

/* DemoSpheres, Scene.cpp:5-12.  Returns the sphere count (6), or -count if cap is smaller.   */
int rt_demo_scene(rt_sphere *out, uint32_t cap);

/* readScene, Utility.cpp:90-160: "camera ox oy oz tx ty tz" / "size N" / N x "sphere rad
 * px py pz ex ey ez cx cy cz mat".  With reference_doubling != 0 the result is what the
 * reference's loader actually hands to the kernel: N value-initialised spheres followed by
 * the N parsed ones (:120,154).  Returns RT_OK and *count, or RT_ERR_ARG (rt_last_error).    */
int rt_read_scene(const char *path, rt_sphere *out, uint32_t cap, uint32_t *count,
                  rt_vec3 *orig, rt_vec3 *target, int reference_doubling);

/* Device-side evaluation of the scalar building blocks, for unit parity tests:
 * op 0: sinf, 1: cosf, 2: pow(x, 1/2.2f), 3: 1/x, 4: sqrt(x), 5: toInt(x) (result as float),
 * 6/7: the kernel's branch-free sinf/cosf (x >= 0).                                           */
int rt_debug_eval(int op, const float *in_host, float *out_host, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* RT_API_H */
