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

/* The library is built with hidden visibility: exactly the functions declared in this header
 * (and, in the diagnostics build librt_hip_diag.so, in rt_debug.h) are exported.               */
#if defined(RT_BUILDING_LIBRARY)
#define RT_API __attribute__((visibility("default")))
#else
#define RT_API
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
    uint64_t sphere_tests;   /* `SphereIntersect` .cl:173 evaluations OF THE REFERENCE'S ALGORITHM for these rays: every closest-hit
                              * query counts all spheres, every shadow query the spheres up to and including its first blocker in
                              * scene order (.cl:215-247).  The plain sweeps execute exactly these tests; the hierarchy of large
                              * scenes (rt_last_kernel "..._pairs") reaches the same answers with far fewer, so there this is the
                              * reference-equivalent count, equal to the oracle's, not the work done: what the walk executes is
                              * counted by the census instance of the diagnostics library (bench.py `roofline.executed`).     */
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
    RT_MODE_FAST   = 1       /* FMA contraction + hardware rcp/rsq/sin/cos/exp2/log2: an approximation of the reference
                              * path whose tolerance is PSNR >= 50 dB against it at equal sample count (north_star).  It
                              * MEETS that on scenes without many small curved mirrors / glass spheres -- the Demo scene at
                              * 1080p x 64 (61.7 dB) and 2160p x 256 (64.3), the 16-sphere scene (58.3) -- and it does NOT on
                              * 256 scattered spheres at 32 spp (46.0), 1024 at 16 spp (37.6) or the 64-sphere mirror box
                              * (27.1): there a last-bit difference in a direction, amplified by curved reflections, flips a
                              * hit decision, and from that sample on the pixel's one sequential random stream (.cl:143-169)
                              * is consumed differently -- the frames then differ at the noise level although both are correct
                              * renderings.  Use RT_MODE_PARITY where the tolerance matters on such scenes.  Measured per
                              * configuration by tools/fast_gate.py (profiles/r05_fast_gate.jsonl), held by
                              * tests/test_gpu_parity.py::test_fast_mode_against_north_star_gate.                        */
};

#define RT_MAX_SPHERES 262144u /* (the hierarchy numbers its leaves of 8 spheres with 15 bits.  Its tables are staged in LDS while
                                 * five workgroups of that size fit a CU -- 31 KiB, about 1 100 spheres; to about 3 200 spheres its
                                 * pairs still are and only the leaves' spheres are read from HBM / L2 ("..._pairs_m"); beyond,
                                 * everything is ("..._pairs_g"); the plain sweep stages its tables while four workgroups fit a CU
                                 * -- 40 KiB, about 2 500 records -- and reads them through the scalar cache beyond ("..._g"))        */

typedef struct rt_ctx rt_ctx;

/* ---- the headline call ------------------------------------------------------------------
 * Equivalent to: a fresh OpenCLConfigBuffer(w,h) [seeds = never-seeded std::rand() stream
 * clamped to >= 2, OpenCLConfig.cpp:676-680] + sceneSetup + `spp` calls of
 * Config::updateRendering() [Config.cpp:73-81], after which `out` holds what getPixels()
 * returns: out[y*w+x] = R | G<<8 | B<<16, row 0 = bottom of the image (.cl:594-596).
 * `cam` carries dir/x/y already computed by the host (Utility.cpp:71-85).
 * `out` is a host buffer of w*h uint32.  Parity mode, device 0, blocking.
 * The library keeps the device state of the last few (w, h) it was called with -- buffers, the
 * device-resident default seed stream, page-locked staging for the read-back -- so that a host
 * which calls rt_render per frame pays for them once; rt_release_cache() frees it.
 * Thread-safe (calls are serialised inside).                                                  */
RT_API int rt_render(const rt_scene *scene, const rt_camera *cam, uint32_t *out, int w, int h, int spp);
RT_API void rt_release_cache(void);

/* ---- progressive interface: one context = one OpenCLConfigBuffer ------------------------ */

/* ctor + allocateBuffer (OpenCLConfig.cpp:398-400, 613-682): device buffers for colours
 * (12 B/px), seeds (8 B/px, initialised to the default stream) and pixels (4 B/px).         */
RT_API int rt_create(rt_ctx **out, int w, int h);

/* one context that renders on `ngpus` devices of this process (HIP devices
 * 0..ngpus-1; rt_create_multi_on names them).  The image is sharded by interleaved row tiles of
 * `tile_rows` rows (tile t -> device t % ngpus; 0 = the default of 8), every device renders its rows
 * on its own stream, and each frame ends with ONE gather to the first device over RCCL
 * (ncclGroupStart; root: ncclRecv x (n-1); others: ncclSend; ncclGroupEnd; ncclUint32), a
 * de-interleave kernel there and one copy to the host.  The receives and the de-interleave run on a SECOND stream of
 * the first device, into one of two receive slots, so that device's render of frame k + 1 overlaps the gather of frame k
 * (asynchronous frames: rt_render_async; rt_throttle(ctx, 0) and the blocking calls wait for the assembled frame).  Every other call of this header works on
 * such a context as on a plain one and means the whole image: rt_render_pass returns all h rows,
 * rt_get_stats sums the devices, rt_read_colors / rt_read_seeds merge them.  Results are bit-identical
 * to a one-device context by construction (pixels are independent); with ngpus = 1 there is nothing to move: no
 * communicator (RCCL is not loaded), no gather, no de-interleave -- the one shard renders straight into the frame.  One kernel instance renders the whole frame: the first shard measures hierarchy
 * against sweep (rt_scene_choice) and the others follow it.
 * STATUS of the n > 1 RCCL branch: it executes on one GPU against a test double of RCCL (tests/rccl_double.cpp, bound through the diagnostics
 * library's rt_debug_set_rccl_library: communicator, the grouped n - 1 receives and n - 1 sends, de-interleave -- on 2, 3 and 8 shards, C4 at
 * full size on 8 -- and every failure site: ncclGroupStart, ncclSend / ncclRecv inside the group, ncclGroupEnd); against librccl on distinct
 * devices over xGMI it has run on no hardware yet (tests/test_gpu_features.py holds that test, skipped below two devices): treat real links as
 * unverified until it has passed on a multi-GPU node.
 * rt_create_multi_on with ONE device listed ngpus times is that one-GPU rehearsal of the path: RCCL refuses two
 * ranks on one device, so the transfers into the root's receive slots are device-to-device copies there; everything
 * else (shards, slots, de-interleave) is unchanged.  A list that mixes repeated and distinct devices is refused
 * (RT_ERR_ARG).  If a gather fails half-way (an RCCL error inside the group or at ncclGroupEnd) the context is marked
 * unusable: every later call on it returns RT_ERR_STATE naming the failure, and only rt_destroy is meaningful -- it polls the
 * context's streams (never a blocking wait) for up to two seconds and frees everything once they have drained; only a context whose streams
 * still hold a transfer after that keeps its device memory and streams (a teardown that returns instead of hanging).    */
RT_API int rt_create_multi(rt_ctx **out, int w, int h, int ngpus);
RT_API int rt_create_multi_on(rt_ctx **out, int w, int h, const int *devices, int ngpus, int tile_rows);
RT_API int rt_shard_count(const rt_ctx *ctx);                     /* 1 for a plain context        */

/* Same, on HIP device `device`, rendering only the row tiles this rank owns: tile t (rows
 * [t*tile_rows, (t+1)*tile_rows)) belongs to rank t % nranks.  tile_rows must be a positive
 * multiple of 8.  The rank's rows are packed in order into a local pixel buffer of
 * rt_local_rows() rows (interleaved row tiles).  nranks = 1 is rt_create().       */
RT_API int rt_create_sharded(rt_ctx **out, int w, int h, int device, int rank, int nranks,
                             int tile_rows);

RT_API void rt_destroy(rt_ctx *ctx);                              /* freeBuffer, :684-717            */

/* sceneSetup (:720-747) + the per-pass sphere upload (:450).  Copies; caller keeps `spheres`.
 * The 44-byte records go to the device as they are and a small kernel there builds the tables the
 * render kernel reads (centre | radius^2, emission | material, colour | radius, and the light list
 * SampleLights walks, with 4*pi*radius^2 -- all in binary32, the reference's own operations).
 * Scenes with 56 and more small spheres also get a hierarchy over them, its shape chosen by surface area (below 1500 such
 * spheres on the host, which then also knows whether walking it will pay; up to 8192 by a second kernel on the same stream, the
 * caller held for microseconds; beyond that built on the host from the records and copied): it changes which instance renders
 * the scene, never the result (rt_last_kernel, rt_scene_choice).  Up to RT_MAX_SPHERES spheres; tables that do not fit LDS are read
 * from HBM / L2.  Ordered after every launch issued on this context; no device-wide synchronisation.  A refused
 * scene (bad arguments, too many spheres) leaves the previous one in place.                                                  */
RT_API int rt_set_scene(rt_ctx *ctx, const rt_sphere *spheres, uint32_t count);

/* Device-resident scene update: replace spheres [first, first+count) of the scene set
 * by rt_set_scene -- moving spheres, changed materials, lights switched on or off -- without a
 * host wait: the records are staged through page-locked memory and copied asynchronously on `hip_stream` (a hipStream_t;
 * NULL = the default stream, as for rt_render_async); launches issued later on this context see the new scene.  The tables and,
 * for large scenes, the hierarchy are rebuilt ONCE, by the device-side kernels on the stream of the next launch, however many
 * updates precede it (a host that moves 200 scattered spheres by 200 calls pays one rebuild, as for one call over the whole
 * range) -- up to 8192 spheres inside the tree by the device's own build by surface area, with no host wait; beyond that the
 * host builds the fixed (halved) shape from its mirror of the records and that launch waits for the previous such build's
 * upload to have left its staging buffer; the choice between hierarchy and sweep is kept.  The sphere count does not change.
 * `spheres` may be reused as soon as the call returns.                                          */
RT_API int rt_update_spheres_async(rt_ctx *ctx, uint32_t first, uint32_t count, const rt_sphere *spheres,
                                   void *hip_stream);

/* updateCamera's result + the per-pass camera upload (:418).  dir/x/y must be filled in.     */
RT_API int rt_set_camera(rt_ctx *ctx, const rt_camera *cam);

RT_API int rt_set_mode(rt_ctx *ctx, int mode);                    /* enum rt_mode; default parity    */

/* Back to pass 0: mCurrentSample = 0, seeds = default stream, counters cleared.              */
RT_API int rt_reset(rt_ctx *ctx);

/* rt_reset() without a host round trip: the next launch starts from a device-resident copy of
 * the default stream (read in place, nothing is copied) and the counters are cleared by a small
 * kernel on `hip_stream`.  The colour plane and the packed pixels are NOT cleared (pass 0 overwrites
 * every pixel, .cl:580-582): until the next launch rt_read_colors / rt_read_pixels still return the
 * previous frame, rt_read_seeds the default stream.  The launch counter and last_kernel_ms restart.
 * Like every call on a context it must not race with other calls on the same context.        */
RT_API int rt_reset_async(rt_ctx *ctx, void *hip_stream);

/* `n_samples` x { setArguments(); execute(); ++mCurrentSample; } (Config.cpp:73-81) as ONE
 * launch that keeps seeds and the running average in registers, then one D2H copy of the
 * pixel buffer into `out_host` (the full image for an unsharded context, the local rows for
 * a sharded one).  out_host may be NULL to skip the copy.  Blocking.
 * Scheduling, never results: launches leave what each tile cost them, and later launches walk the tiles heaviest first (sorted
 * again from the last costs whenever the scene or the camera has changed).  A long launch (8 passes and more) prices the tiles by
 * itself; a launch of 24 passes or more that has no costs to go by renders 4 of its passes first to get them, then the rest
 * heaviest first -- so a scene's first frame is a few percent slower than its later ones.  Short launches -- a pass per call, the
 * reference's own regime -- add their costs up instead, and the one that finds 16 passes' worth sorts from them: a host that only
 * ever launches a pass or two at a time gets the same order (8-23 % on scenes of hundreds of spheres and more) from its 17th pass on.  */
RT_API int rt_render_pass(rt_ctx *ctx, uint32_t *out_host, int n_samples);

/* Page-lock the host buffer that rt_render_pass copies into (the host's `pPixels`,
 * OpenCLConfig.cpp:618-621), so the per-pass readback of the reference's display loop
 * (clEnqueueReadBuffer after every pass, OpenCLConfig.cpp:498-512) runs at the full PCIe rate
 * instead of through a pageable staging copy.  `count` uint32 from `out_host` must stay valid
 * and at the same address until rt_pin_output(ctx, NULL, 0) or rt_destroy.  Optional.        */
RT_API int rt_pin_output(rt_ctx *ctx, uint32_t *out_host, size_t count);

/* enable = 0: later launches advance seeds and the running average but leave the packed pixel
 * buffer alone (no toInt, .cl:34,594-596, and no pixel store) -- for passes whose frame nobody
 * will look at; the next launch with enable = 1 writes every pixel of its frame from the running
 * average, so nothing is lost.  Default 1.                                                    */
RT_API int rt_set_pixel_write(rt_ctx *ctx, int enable);

/* getPixels() for hosts that skip pixel stores: brings the packed frame up to date -- if the last
 * launches ran with the pixel store off, a small kernel packs the frame from the running average
 * (same toInt, .cl:34,594-596) -- and copies it to `out_host` (local rows x w uint32).  Waits for
 * the launches issued on this context.                                                        */
RT_API int rt_read_pixels(rt_ctx *ctx, uint32_t *out_host);

/* Same launch, asynchronous on `hip_stream` (a hipStream_t, NULL = default stream), no copy
 * and no synchronisation: the caller orders later work on that stream.                       */
RT_API int rt_render_async(rt_ctx *ctx, int n_samples, void *hip_stream);

/* rt_read_pixels without the wait: the packed frame is brought up to date and its copy to `out_host` is queued on
 * `hip_stream` behind everything issued on this context; the buffer holds the frame once that stream has drained (or
 * rt_throttle(ctx, 0, ...) has returned).  `out_host` should be page-locked (rt_pin_output), or the runtime stages
 * the copy and the call may block.  On a multi-device context this is rt_read_pixels (it blocks).               */
RT_API int rt_read_pixels_async(rt_ctx *ctx, uint32_t *out_host, void *hip_stream);

/* Pacing for hosts that queue passes with rt_render_async (the adapter's display loop): returns when at most
 * `max_in_flight` of the rt_render_async launches issued on this context since the first rt_throttle call are still
 * unfinished (0 = wait for all of them).  *ms_per_pass (may be NULL) receives the device time per pass of the most
 * recent launch that has finished, 0 if none has.  The first call only switches the bookkeeping on (two events per
 * launch from then on) and returns at once.  A multi-device context waits for everything when max_in_flight is 0
 * and reports no time.                                                                                          */
RT_API int rt_throttle(rt_ctx *ctx, int max_in_flight, double *ms_per_pass);

/* The context's own non-blocking stream (a hipStream_t), the one rt_render_pass uses.  With
 * several contexts in flight, rt_render_async(ctx, n, rt_stream(ctx)) runs each on its own.
 * HIP gives a process GPU_MAX_HW_QUEUES hardware queues (default 4) and lets further streams
 * share them: two contexts on one queue do not overlap at all, so a host that keeps F contexts
 * in flight should start with GPU_MAX_HW_QUEUES >= the number of streams it uses (bench.py: 8).
 * PLATFORM PRECONDITION (profiles/r02_stale_seed/): do not put more hardware queues on ONE GPU
 * than its scheduler keeps resident -- one process per GPU is always fine; four processes of 8 queues each on one
 * GPU were not: the scheduler then time-slices the queues, a dispatch can run XCD-share by XCD-share around a
 * descheduling, and on this driver stack the early share occasionally loses its writes (no fence or store form the
 * library could issue prevents it).  tools/gather_stress.py is the acceptance test for a deployment.
 * Of a MULTI-DEVICE context: the stream its assembled frame is complete on -- with several devices the first device's
 * gather stream, behind the receives and the de-interleave of the last frame queued (rt_render_async takes no stream there:
 * every device renders on its own).  Work queued on it after rt_render_async sees the whole frame in rt_device_pixels, and
 * the next frame's assembly waits for that work: this is the stream that orders the frame buffer.                    */
RT_API void *rt_stream(rt_ctx *ctx);

/* Device address and element count (uint32) of the local pixel buffer.                       */
RT_API int rt_device_pixels(rt_ctx *ctx, void **dptr, size_t *count);

/* Redirect the packed pixels of later launches into a caller-owned DEVICE buffer of at least
 * rt_local_rows()*w uint32 (e.g. the send buffer of the frame-end gather, so no copy is needed);
 * NULL restores the context's own buffer.  The caller keeps the buffer alive and orders its
 * reuse against the launches it issued.                                                      */
RT_API int rt_set_pixel_buffer(rt_ctx *ctx, void *dptr, size_t count);

RT_API int rt_local_rows(const rt_ctx *ctx);                      /* rows this context renders       */
RT_API int rt_current_sample(const rt_ctx *ctx);                  /* mCurrentSample                  */

/* Copies of the reference's other two buffers, for parity checks: the colour plane
 * (3 floats/px, y-flipped as .cl:579 stores it) and the seed pairs, full image size; rows
 * this rank does not own keep their initial content.                                         */
RT_API int rt_read_colors(rt_ctx *ctx, float *out_host);
RT_API int rt_read_seeds(rt_ctx *ctx, uint32_t *out_host);

RT_API int rt_get_stats(rt_ctx *ctx, rt_stats *out);
/* The kernel instance the context's last launch used, by its symbol (what a profiler lists): the library picks it
 * from the scene -- "rt_trace_parity_w1" (few spheres: one wavefront per workgroup), "..._coop_w1" / "..._coop"
 * (12 and more: wave-ballot any-hit sharing; with 4 to 11 spheres whichever of the two the scene's first launches timed
 * faster -- coop warm, coop timed, plain warm, plain timed: passes of the frame like any other), "..._pairs" (hundreds of
 * small spheres: a hierarchy, where it measured faster than the sweep on this scene; "..._pairs_m" / "..._pairs_g" when its
 * tables outgrow LDS), "..._g" (a plain sweep over a table beyond its LDS budget, read through the scalar cache), the same with "fast".  "" before the first launch.
 * Frames do not depend on it. */
RT_API const char *rt_last_kernel(const rt_ctx *ctx);
/* Hierarchy or plain sweep for the current scene (scenes with 56 to 1500 small spheres; larger ones always walk the
 * hierarchy).  rt_set_scene builds the tree on the host and with it a surface-area estimate of what a ray costs either
 * way; where the estimate is clear (predicted ratio outside 0.75 .. 1.33) it decides and nothing is measured, so a new
 * scene's first frame costs what a frame costs.  Inside that band -- and after device-resident updates that changed the
 * tree's size by a quarter -- the first launches time both forms (hierarchy warm, hierarchy timed, sweep warm, sweep timed;
 * passes of the frame like any other) and the faster one renders the rest.  Returns 0 = not decided (yet, or a scene that
 * has no choice), 1 = hierarchy, 2 = plain sweep, and the two MEASURED times per pass in milliseconds (0 when the estimate
 * decided or nothing was measured).  Never blocks; of a multi-device context, the first shard's. */
RT_API int rt_scene_choice(rt_ctx *ctx, double *hierarchy_ms_per_pass, double *sweep_ms_per_pass);

/* Identity of the sources and compiler flags this library was built from (16 hex digits).  tools/summarize_profile.py stamps
 * every profile record with it; bench.py prints a committed profile's counters only beside the library they were measured on. */
RT_API const char *rt_build_id(void);

/* Text of the calling thread's last failure ("" if none).                                    */
RT_API const char *rt_last_error(void);

/* ---- host-side helpers either side of the path ---------------------------- */

/* computeCameraVariables, Utility.cpp:71-85 (Vec::norm's double sqrt, Vec.cpp:28-30).        */
RT_API void rt_compute_camera(rt_camera *cam, int w, int h);

/* The seed initialisation of OpenCLConfig.cpp:676-680 without depending on the host libc:
 * glibc's never-seeded rand() stream restated, each value clamped to >= 2.                   */
RT_API void rt_default_seeds(uint32_t *seeds, size_t count);

/* DemoSpheres, Scene.cpp:5-12.  Returns the sphere count (6), or -count if cap is smaller.   */
RT_API int rt_demo_scene(rt_sphere *out, uint32_t cap);

/* readScene, Utility.cpp:90-160: "camera ox oy oz tx ty tz" / "size N" / N x "sphere rad
 * px py pz ex ey ez cx cy cz mat".  With reference_doubling != 0 the result is what the
 * reference's loader actually hands to the kernel: N value-initialised spheres followed by
 * the N parsed ones (:120,154).  Returns RT_OK and *count, or RT_ERR_ARG (rt_last_error).    */
RT_API int rt_read_scene(const char *path, rt_sphere *out, uint32_t cap, uint32_t *count,
                  rt_vec3 *orig, rt_vec3 *target, int reference_doubling);

/* ---- frame assembly on the gather root --------------------------------------
 * De-interleave kernel: `gathered` holds the ranks' local pixel blocks one after another, rank r
 * at gathered + r * pad_rows * w (its rows packed in order, as rt_create_sharded lays them out);
 * full[y * w + x] = the pixel of image row y.  Both are DEVICE pointers on `device`; asynchronous
 * on `hip_stream`.  Used by the in-library multi-GPU context and by process-per-GPU hosts after
 * their own gather (bench.py: torch.distributed over RCCL).                                    */
RT_API int rt_deinterleave_rows(uint32_t *full, const uint32_t *gathered, int w, int h, int nranks,
                                int tile_rows, int pad_rows, int device, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* RT_API_H */
