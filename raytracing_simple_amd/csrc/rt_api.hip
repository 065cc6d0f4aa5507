// rt_api.hip -- the C ABI of include/rt_api.h: context = the reference's OpenCLConfigBuffer
// (SimpleRT/src/OpenCLConfig.cpp:398-747) re-done for one MI355X: device buffers, scene
// tables, launch geometry, row-tile sharding, counters.  No CPU fallback: without a HIP
// device every entry point fails with RT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "rt_device.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(call)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(RT_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                \
    } while (0)

}  // namespace

// rt_reset_async zeroes the work counters with a kernel on the caller's stream and restores NO seeds:
// the next launch reads the pristine default stream directly (LaunchParams::seeds_in).  Reason: in
// the multi-rank frame loop (tools/gather_stress.py: 4 processes sharing one GPU, 6 streams each, a
// torch.distributed collective per frame) about one frame in a thousand was rendered from seeds that
// the reset issued just before it ON THE SAME STREAM -- a device-to-device copy at first, then a copy
// kernel -- had not restored yet: every wrong pixel equalled the frame computed from un-reset seeds,
// and a probe kernel placed between reset and launch (RT_PROBE=1) saw whole workgroups' worth of
// un-restored seed words.  It needs the collective's worker in the process: the same loop without
// the gather, with several host threads, or as pure HIP (tools/ubench/stream_order.hip: dependent
// kernels, events, waits, a copying worker thread, 4-6 processes) never showed it, so the cause is
// not pinned down.  A frame that starts from data nobody writes cannot lose that race.
__global__ void rt_zero_counters_kernel(unsigned long long *counters, unsigned long long *stats) {
    for (int i = threadIdx.x; i < 32; i += blockDim.x) counters[i] = 0ull;
    for (int i = threadIdx.x; i < rt::kStatReplicas * 8; i += blockDim.x) stats[i] = 0ull;
}

// diagnostic only (rt_debug_reset_by_copy): the reset this library used before -- a copy kernel that
// restores the seed words, which the next launch then reads back
__global__ void rt_debug_copy_seeds_kernel(uint32_t *seeds, const uint32_t *seeds0, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) seeds[i] = seeds0[i];
}

__global__ void rt_debug_probe_seeds_kernel(const uint32_t *seeds, const uint32_t *seeds0, size_t n, unsigned long long *out) {
    unsigned long long b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b += (seeds[i] != seeds0[i]);
    if (b) atomicAdd(out, b);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(out + 1, 1ull);     // probes run
}

struct rt_ctx {
    int device = 0;
    int w = 0, h = 0;
    int rank = 0, nranks = 1, tile_rows = 8, local_rows = 0;
    uint32_t *d_seeds = nullptr;
    uint32_t *d_seeds0 = nullptr;  // pristine default stream, for device-side resets
    float *d_colors = nullptr;
    uint32_t *d_pixels = nullptr;
    uint32_t *d_pixels_ext = nullptr;   // caller-owned target of rt_set_pixel_buffer, or null
    void *pinned_out = nullptr;         // host buffer page-locked by rt_pin_output, or null
    int pixel_write = 1;                // rt_set_pixel_write
    bool seeds_default = false;         // after rt_reset_async: the next launch reads the pristine stream
    unsigned long long *d_counters = nullptr;
    unsigned long long *d_stats = nullptr;      // rt::kStatReplicas x 8 partial work counters
    float4 *d_tables = nullptr;   // geom | emis | colr | lightA | lightB, one allocation
    size_t tables_cap = 0;        // in float4
    rt::SceneTables scene{};
    rt_camera cam{};
    bool have_scene = false, have_cam = false;
    int mode = RT_MODE_PARITY;
    int regen_gate = 0;           // 0 = choose from the scene size
    int mat_lds_limit = 24 * 1024;
    int coop_min = 12;
    int persist = 0;              // persistent-wavefront instances (tile queue + per-lane pixel hand-out)
    int n_cus = 256;
            // scenes with at least this many spheres use the cooperative any-hit instance (0 = never)
    int current_sample = 0;
    uint64_t launches = 0;
    double last_ms = 0.0;
    unsigned long long debug_counters[24] = {};   // diagnostic instances only
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

int select_device(const rt_ctx *c) {
    HIP_TRY(hipSetDevice(c->device));
    return RT_OK;
}

// Everything the context does to its own buffers goes through its own stream: that stream is
// non-blocking, so work put on the null stream (hipMemset, device-to-device hipMemcpy: both return
// before they have run) would not be ordered against the launches that follow -- under load from
// other host threads a frame could start on seeds and counters that were still being reset.
int upload_default_seeds(rt_ctx *c) {
    const size_t count = 2 * (size_t)c->w * (size_t)c->h;
    std::vector<uint32_t> host(count);
    rt_default_seeds(host.data(), count);
    HIP_TRY(hipMemcpyAsync(c->d_seeds0, host.data(), count * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_seeds, c->d_seeds0, count * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_colors, 0, 3 * (size_t)c->w * (size_t)c->h * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_pixels, 0, (size_t)c->local_rows * (size_t)c->w * sizeof(uint32_t), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, 32 * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_stats, 0, rt::kStatReplicas * 8 * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));          // `host` goes out of scope; rt_create / rt_reset are blocking calls
    return RT_OK;
}

int launch(rt_ctx *c, int n_samples, hipStream_t stream) {
    if (!c->have_scene || !c->have_cam)
        return fail(RT_ERR_STATE, "rt_set_scene and rt_set_camera must precede rendering");
    if (n_samples < 0) return fail(RT_ERR_ARG, "n_samples < 0");
    if (n_samples > 0x7fffffff - c->current_sample)
        return fail(RT_ERR_ARG, "pass counter would overflow (%d + %d)", c->current_sample, n_samples);
    if (n_samples == 0 || c->local_rows == 0) return RT_OK;

    rt::LaunchParams p{};
    p.scene = c->scene;
    p.cam = c->cam;
    p.seeds = c->d_seeds;
    p.seeds_in = c->seeds_default ? c->d_seeds0 : c->d_seeds;
    p.colors = c->d_colors;
    p.pixels = c->d_pixels_ext ? c->d_pixels_ext : c->d_pixels;
    p.counters = c->d_counters;
    p.stats = c->d_stats;
    p.w = c->w;
    p.h = c->h;
    p.first_sample = c->current_sample;
    p.n_samples = n_samples;
    p.rank = c->rank;
    p.nranks = c->nranks;
    p.tile_rows = c->tile_rows;
    p.local_rows = c->local_rows;
    p.skip_pixels = c->pixel_write ? 0 : 1;
    p.inv_w = 1.f / (float)c->w;          // correctly rounded on the host as on the device (-ffp-contract=off, IEEE division)
    p.inv_h = 1.f / (float)c->h;
    p.regen_gate = c->regen_gate > 0 ? c->regen_gate : (c->scene.n_spheres <= 512 ? 8 : 1);
    const size_t lds_all = rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, true, n_samples);
    // materials ride along in LDS only while that keeps at least 6 workgroups per CU resident
    // (160 KiB / 24 KiB); larger scenes read them from L2 once per hit
    p.mat_in_lds = lds_all <= (size_t)c->mat_lds_limit;
    const size_t lds = rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, p.mat_in_lds != 0, n_samples);

    dim3 grid((unsigned)((c->w + rt::kTileW - 1) / rt::kTileW),
              (unsigned)((c->local_rows + rt::kTileH - 1) / rt::kTileH));
    hipError_t e;
    const bool coop = c->coop_min > 0 && c->scene.n_spheres >= (uint32_t)c->coop_min;
    const bool persist = c->persist != 0 && (c->mode == RT_MODE_FAST || c->mode == RT_MODE_PARITY);
    p.tiles_x = (c->w + 7) / 8;
    p.n_tiles = p.tiles_x * ((c->local_rows + 7) / 8);
    if (persist) {
        // just enough workgroups to fill the machine; the tile queue (counters[30]) does the rest
        size_t per_cu = lds > 0 ? (160 * 1024) / (lds + 6 * 1024) : 6;
        if (per_cu > 6) per_cu = 6;
        if (per_cu < 1) per_cu = 1;
        size_t blocks = (size_t)c->n_cus * per_cu;
        const size_t needed = ((size_t)p.n_tiles + 3) / 4;
        if (blocks > needed) blocks = needed;
        grid = dim3((unsigned)blocks, 1, 1);
        HIP_TRY(hipMemsetAsync(c->d_counters + 30, 0, sizeof(unsigned long long), stream));
    }
    if (persist && c->mode == RT_MODE_FAST)
        e = rt::launch_fast(coop ? rt::kFastPersistCoopVariant : rt::kFastPersistVariant, p, grid, lds, stream);
    else if (persist)
        e = rt::launch_parity(coop ? rt::kParityPersistCoopVariant : rt::kParityPersistVariant, p, grid, lds, stream);
    else if (c->mode == RT_MODE_FAST) e = rt::launch_fast(coop ? rt::kFastCoopVariant : 0, p, grid, lds, stream);
    else if (c->mode >= 200) e = rt::launch_fast(c->mode - 200, p, grid, lds, stream);
    else if (c->mode >= 100) e = rt::launch_parity(c->mode - 100, p, grid, lds, stream);
    else e = rt::launch_parity(coop ? rt::kParityCoopVariant : 0, p, grid, lds, stream);
    if (e != hipSuccess)
        return fail(RT_ERR_HIP, "kernel launch failed: %s (grid %ux%u, lds %zu B)",
                    hipGetErrorString(e), grid.x, grid.y, lds);
    c->current_sample += n_samples;
    c->launches += 1;
    c->seeds_default = false;           // this launch has written every seed pair the context renders
    return RT_OK;
}

}  // namespace

extern "C" {

const char *rt_last_error(void) { return g_err; }

// used by rt_host.cpp (not part of the public header)
void rt_host_set_error(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg ? msg : ""); }

int rt_create_sharded(rt_ctx **out, int w, int h, int device, int rank, int nranks,
                      int tile_rows) {
    if (!out) return fail(RT_ERR_ARG, "out is null");
    *out = nullptr;
    if (w <= 0 || h <= 0) return fail(RT_ERR_ARG, "image size %dx%d", w, h);
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(RT_ERR_ARG, "rank %d of %d", rank, nranks);
    if (tile_rows <= 0 || tile_rows % rt::kTileH != 0)
        return fail(RT_ERR_ARG, "tile_rows must be a positive multiple of %d", rt::kTileH);

    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(RT_ERR_NO_DEVICE, "no HIP device (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= n_dev) return fail(RT_ERR_ARG, "device %d of %d", device, n_dev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const int n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RT_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device,
                    prop.gcnArchName);

    rt_ctx *c = new (std::nothrow) rt_ctx();
    if (!c) return fail(RT_ERR_ALLOC, "host allocation failed");
    c->device = device;
    c->n_cus = n_cus;
    c->w = w;
    c->h = h;
    c->rank = rank;
    c->nranks = nranks;
    c->tile_rows = tile_rows;
    const int n_tiles = (h + tile_rows - 1) / tile_rows;
    int rows = 0;
    for (int t = rank; t < n_tiles; t += nranks) {
        const int r0 = t * tile_rows;
        rows += (r0 + tile_rows <= h) ? tile_rows : (h - r0);
    }
    c->local_rows = rows;

    int rc = select_device(c);
    const size_t px = (size_t)w * (size_t)h;
    auto alloc_all = [&]() -> int {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&c->ev0));
        HIP_TRY(hipEventCreate(&c->ev1));
        HIP_TRY(hipMalloc(&c->d_seeds, 2 * px * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->d_seeds0, 2 * px * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->d_colors, 3 * px * sizeof(float)));
        HIP_TRY(hipMalloc(&c->d_pixels, ((size_t)rows * w + 1) * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->d_counters, 32 * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc(&c->d_stats, rt::kStatReplicas * 8 * sizeof(unsigned long long)));
        HIP_TRY(rt::prepare_parity());
        HIP_TRY(rt::prepare_fast());
        return upload_default_seeds(c);
    };
    if (rc == RT_OK) rc = alloc_all();
    if (rc != RT_OK) {
        rt_destroy(c);
        return rc;
    }
    *out = c;
    return RT_OK;
}

int rt_create(rt_ctx **out, int w, int h) { return rt_create_sharded(out, w, h, 0, 0, 1, rt::kTileH); }

void rt_destroy(rt_ctx *c) {
    if (!c) return;
    if (hipSetDevice(c->device) == hipSuccess) {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (c->pinned_out) (void)hipHostUnregister(c->pinned_out);
        (void)hipFree(c->d_seeds);
        (void)hipFree(c->d_seeds0);
        (void)hipFree(c->d_colors);
        (void)hipFree(c->d_pixels);
        (void)hipFree(c->d_counters);
        (void)hipFree(c->d_stats);
        (void)hipFree(c->d_tables);
        if (c->ev0) (void)hipEventDestroy(c->ev0);
        if (c->ev1) (void)hipEventDestroy(c->ev1);
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

int rt_set_scene(rt_ctx *c, const rt_sphere *spheres, uint32_t count) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    if (count > 0 && !spheres) return fail(RT_ERR_ARG, "spheres is null");
    if (count > RT_MAX_SPHERES) return fail(RT_ERR_ARG, "%u spheres > RT_MAX_SPHERES (%u)", count, RT_MAX_SPHERES);
    int rc = select_device(c);
    if (rc != RT_OK) return rc;

    // host-side table build: strict binary32 (this file is compiled -ffp-contract=off)
    std::vector<float4> geom(count), emis(count), colr(count), la, lb;
    for (uint32_t i = 0; i < count; ++i) {
        const rt_sphere &s = spheres[i];
        geom[i] = make_float4(s.p.x, s.p.y, s.p.z, s.rad * s.rad);                     // .cl:184
        float refl_bits;
        int32_t refl = s.refl;
        memcpy(&refl_bits, &refl, 4);
        emis[i] = make_float4(s.e.x, s.e.y, s.e.z, refl_bits);
        colr[i] = make_float4(s.c.x, s.c.y, s.c.z, s.rad);
        if (!((s.e.x == 0.f) && (s.e.z == 0.f))) {                                     // .cl:135-138,266
            la.push_back(make_float4(s.p.x, s.p.y, s.p.z, s.rad));
            lb.push_back(make_float4(s.e.x, s.e.y, s.e.z,
                                     4.f * 3.14159265358979323846f * s.rad * s.rad));  // .cl:297
        }
    }
    const uint32_t nl = (uint32_t)la.size();
    const size_t need = 3 * (size_t)count + 2 * (size_t)nl + 1;
    // ordered after any launch still reading the old tables
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipDeviceSynchronize());
    if (need > c->tables_cap) {
        (void)hipFree(c->d_tables);
        c->d_tables = nullptr;
        c->tables_cap = 0;
        HIP_TRY(hipMalloc(&c->d_tables, need * sizeof(float4)));
        c->tables_cap = need;
    }
    float4 *base = c->d_tables;
    float4 *d_geom = base, *d_emis = base + count, *d_colr = base + 2 * (size_t)count;
    float4 *d_la = base + 3 * (size_t)count, *d_lb = d_la + nl;
    if (count) {
        HIP_TRY(hipMemcpy(d_geom, geom.data(), count * sizeof(float4), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_emis, emis.data(), count * sizeof(float4), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_colr, colr.data(), count * sizeof(float4), hipMemcpyHostToDevice));
    }
    if (nl) {
        HIP_TRY(hipMemcpy(d_la, la.data(), nl * sizeof(float4), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_lb, lb.data(), nl * sizeof(float4), hipMemcpyHostToDevice));
    }
    c->scene = rt::SceneTables{ d_geom, d_emis, d_colr, d_la, d_lb, count, nl };
    if (rt::lds_bytes(count, nl, false) > 152 * 1024)
        return fail(RT_ERR_ARG, "scene needs %zu B of LDS (> 152 KiB)", rt::lds_bytes(count, nl, false));
    c->have_scene = true;
    return RT_OK;
}

int rt_set_camera(rt_ctx *c, const rt_camera *cam) {
    if (!c || !cam) return fail(RT_ERR_ARG, "null argument");
    c->cam = *cam;
    c->have_cam = true;
    return RT_OK;
}

int rt_set_mode(rt_ctx *c, int mode) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    // 100+k / 200+k: A/B instances of the parity / fast arithmetic (not part of the contract)
    const bool ab = (mode >= 100 && mode < 100 + rt::parity_variant_count()) ||
                    (mode >= 200 && mode < 200 + rt::fast_variant_count());
    if (mode != RT_MODE_PARITY && mode != RT_MODE_FAST && !ab) return fail(RT_ERR_ARG, "mode %d", mode);
    c->mode = mode;
    return RT_OK;
}

int rt_reset(rt_ctx *c) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipDeviceSynchronize());
    c->current_sample = 0;
    c->launches = 0;
    c->last_ms = 0.0;
    c->seeds_default = false;
    return upload_default_seeds(c);
}

int rt_reset_async(rt_ctx *c, void *hip_stream) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    hipLaunchKernelGGL(rt_zero_counters_kernel, dim3(1), dim3(256), 0, (hipStream_t)hip_stream, c->d_counters, c->d_stats);
    HIP_TRY(hipGetLastError());
    c->seeds_default = true;            // the next launch reads d_seeds0
    c->current_sample = 0;
    return RT_OK;
}

int rt_debug_reset_by_copy(rt_ctx *c, void *hip_stream, int use_memcpy) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    const size_t n = 2 * (size_t)c->w * (size_t)c->h;
    if (use_memcpy) {
        HIP_TRY(hipMemcpyAsync(c->d_seeds, c->d_seeds0, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
    } else {
        hipLaunchKernelGGL(rt_debug_copy_seeds_kernel, dim3(1024), dim3(256), 0, (hipStream_t)hip_stream, c->d_seeds, c->d_seeds0, n);
        HIP_TRY(hipGetLastError());
    }
    c->seeds_default = false;
    c->current_sample = 0;
    return RT_OK;
}

// diagnostic: a kernel on `hip_stream` that counts the seed words differing from the default stream
// into counters[28] (and the number of probes into counters[29]); read them with rt_debug_counters_raw
int rt_debug_probe_seeds(rt_ctx *c, void *hip_stream) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    hipLaunchKernelGGL(rt_debug_probe_seeds_kernel, dim3(256), dim3(256), 0, (hipStream_t)hip_stream, c->d_seeds, c->d_seeds0,
                       2 * (size_t)c->w * (size_t)c->h, c->d_counters + 28);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_debug_counters_raw(rt_ctx *c, unsigned long long *out32) {
    if (!c || !out32) return fail(RT_ERR_ARG, "null argument");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out32, c->d_counters, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_render_async(rt_ctx *c, int n_samples, void *hip_stream) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    return launch(c, n_samples, (hipStream_t)hip_stream);
}

int rt_render_pass(rt_ctx *c, uint32_t *out_host, int n_samples) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    rc = launch(c, n_samples, c->stream);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    // the launch is complete before the readback is issued (not merely queued behind it: see
    // rt_reset_kernel for why this library does not lean on copy-after-kernel ordering)
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (out_host && c->local_rows > 0) {
        HIP_TRY(hipMemcpyAsync(out_host, c->d_pixels_ext ? c->d_pixels_ext : c->d_pixels,
                               (size_t)c->local_rows * c->w * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_ms = ms;
    return RT_OK;
}

int rt_set_pixel_write(rt_ctx *c, int enable) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    c->pixel_write = enable ? 1 : 0;
    return RT_OK;
}

int rt_pin_output(rt_ctx *c, uint32_t *out_host, size_t count) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    if (c->pinned_out) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipHostUnregister(c->pinned_out);
        c->pinned_out = nullptr;
    }
    if (!out_host || (size_t)c->local_rows * (size_t)c->w == 0) return RT_OK;    // nothing to pin (a rank without rows)
    if (count < (size_t)c->local_rows * (size_t)c->w)
        return fail(RT_ERR_ARG, "output buffer of %zu < %zu elements", count, (size_t)c->local_rows * (size_t)c->w);
    HIP_TRY(hipHostRegister(out_host, count * sizeof(uint32_t), hipHostRegisterDefault));
    c->pinned_out = out_host;
    return RT_OK;
}

int rt_set_pixel_buffer(rt_ctx *c, void *dptr, size_t count) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    if (dptr && count < (size_t)c->local_rows * (size_t)c->w)
        return fail(RT_ERR_ARG, "pixel buffer of %zu < %zu elements", count, (size_t)c->local_rows * (size_t)c->w);
    c->d_pixels_ext = static_cast<uint32_t *>(dptr);
    return RT_OK;
}

// The context's own stream (hipStream_t, non-blocking): what rt_render_pass launches on.  Callers
// that keep several contexts in flight can launch each on its own stream through
// rt_render_async(ctx, n, rt_stream(ctx)).
void *rt_stream(rt_ctx *c) { return c ? (void *)c->stream : nullptr; }

int rt_device_pixels(rt_ctx *c, void **dptr, size_t *count) {
    if (!c || !dptr || !count) return fail(RT_ERR_ARG, "null argument");
    *dptr = c->d_pixels_ext ? c->d_pixels_ext : c->d_pixels;
    *count = (size_t)c->local_rows * (size_t)c->w;
    return RT_OK;
}

int rt_local_rows(const rt_ctx *c) { return c ? c->local_rows : RT_ERR_ARG; }
int rt_current_sample(const rt_ctx *c) { return c ? c->current_sample : RT_ERR_ARG; }

int rt_read_colors(rt_ctx *c, float *out_host) {
    if (!c || !out_host) return fail(RT_ERR_ARG, "null argument");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());   // also covers rt_render_async on a caller's stream
    HIP_TRY(hipDeviceSynchronize());   // also covers rt_render_async on a caller's stream
    HIP_TRY(hipMemcpyAsync(out_host, c->d_colors, 3 * (size_t)c->w * c->h * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_read_seeds(rt_ctx *c, uint32_t *out_host) {
    if (!c || !out_host) return fail(RT_ERR_ARG, "null argument");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());   // also covers rt_render_async on a caller's stream
    HIP_TRY(hipMemcpyAsync(out_host, c->seeds_default ? c->d_seeds0 : c->d_seeds, 2 * (size_t)c->w * c->h * sizeof(uint32_t),
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_get_stats(rt_ctx *c, rt_stats *out) {
    if (!c || !out) return fail(RT_ERR_ARG, "null argument");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    unsigned long long v[32];
    HIP_TRY(hipDeviceSynchronize());   // also covers rt_render_async on a caller's stream
    HIP_TRY(hipMemcpyAsync(v, c->d_counters, sizeof v, hipMemcpyDeviceToHost, c->stream));
    unsigned long long part[rt::kStatReplicas * 8], sum[5] = { 0, 0, 0, 0, 0 };
    HIP_TRY(hipMemcpyAsync(part, c->d_stats, sizeof part, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int r = 0; r < rt::kStatReplicas; ++r)
        for (int k = 0; k < 5; ++k) sum[k] += part[r * 8 + k];
    out->samples = sum[0];
    out->closest_rays = sum[1];
    out->shadow_rays = sum[2];
    out->sphere_tests = sum[3];
    out->rng_draws = sum[4];
    memcpy(c->debug_counters, v + 8, sizeof c->debug_counters);
    out->launches = c->launches;
    out->last_kernel_ms = c->last_ms;
    return RT_OK;
}

// tuning knob (not part of the contract): 0 = automatic, 1 = free-running, n = gate of n lanes
int rt_debug_set_regen_gate(rt_ctx *c, int gate) {
    if (!c || gate < 0 || gate > 64) return fail(RT_ERR_ARG, "gate %d", gate);
    c->regen_gate = gate;
    return RT_OK;
}

int rt_debug_set_mat_lds_limit(rt_ctx *c, int bytes) {
    if (!c || bytes < 0) return fail(RT_ERR_ARG, "bytes %d", bytes);
    c->mat_lds_limit = bytes;
    return RT_OK;
}

int rt_debug_set_persist(rt_ctx *c, int on) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    c->persist = on ? 1 : 0;
    return RT_OK;
}

int rt_debug_set_ncus(rt_ctx *c, int n) {      // shrink the persistent grid (tests of the tile queue)
    if (!c || n < 1) return fail(RT_ERR_ARG, "n %d", n);
    c->n_cus = n;
    return RT_OK;
}

int rt_debug_set_coop_min(rt_ctx *c, int min_spheres) {
    if (!c || min_spheres < 0) return fail(RT_ERR_ARG, "min_spheres %d", min_spheres);
    c->coop_min = min_spheres;
    return RT_OK;
}

// diagnostic: section cycle sums of a stamped instance (valid after rt_get_stats)
int rt_debug_counters(rt_ctx *c, unsigned long long *out24) {
    if (!c || !out24) return fail(RT_ERR_ARG, "null argument");
    memcpy(out24, c->debug_counters, sizeof c->debug_counters);
    return RT_OK;
}

int rt_render(const rt_scene *scene, const rt_camera *cam, uint32_t *out, int w, int h, int spp) {
    if (!scene || !cam || !out) return fail(RT_ERR_ARG, "null argument");
    if (spp < 0) return fail(RT_ERR_ARG, "spp < 0");
    rt_ctx *c = nullptr;
    int rc = rt_create(&c, w, h);
    if (rc == RT_OK) rc = rt_set_scene(c, scene->spheres, scene->count);
    if (rc == RT_OK) rc = rt_set_camera(c, cam);
    if (rc == RT_OK) rc = rt_render_pass(c, out, spp);
    rt_destroy(c);
    return rc;
}

// exhaustive device-side check of the lean correctly-rounded sqrt: mismatches over all 2^32 inputs
static long long sqrt_check(int which) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device");
    unsigned long long *d = nullptr, h = 0;
    if (hipSetDevice(0) != hipSuccess || hipMalloc(&d, 8) != hipSuccess) return fail(RT_ERR_HIP, "alloc");
    hipError_t e = hipMemset(d, 0, 8);
    if (e == hipSuccess) e = rt::launch_sqrt_check(d, nullptr, which);
    if (e == hipSuccess) e = hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "sqrt check %d: %s", which, hipGetErrorString(e));
    return (long long)h;
}
long long rt_debug_sqrt_mismatches(void) { return sqrt_check(0); }
// sphere test with the unchecked square root against the one with sqrtf, tiny discriminants
long long rt_debug_hitpost_mismatches(void) { return sqrt_check(1); }

// mismatches of the candidate lean reciprocals per input exponent: out[4][256]
int rt_debug_rcp_probe(unsigned long long *out1024) {
    if (!out1024) return fail(RT_ERR_ARG, "null argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device");
    unsigned long long *d = nullptr;
    if (hipSetDevice(0) != hipSuccess || hipMalloc(&d, 8192) != hipSuccess) return fail(RT_ERR_HIP, "alloc");
    hipError_t e = hipMemset(d, 0, 8192);
    if (e == hipSuccess) e = rt::launch_rcp_probe(d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out1024, d, 8192, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_debug_rcp_probe: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_debug_eval(int op, const float *in_host, float *out_host, size_t n) {
    if ((!in_host || !out_host) && n) return fail(RT_ERR_ARG, "null argument");
    if (op < 0 || op > 8) return fail(RT_ERR_ARG, "op %d", op);
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device");
    if (n == 0) return RT_OK;
    float *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(hipSetDevice(0));
    HIP_TRY(hipMalloc(&d_in, n * sizeof(float)));
    hipError_t e = hipMalloc(&d_out, n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_in, in_host, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rt::launch_eval_parity(op, d_in, d_out, n, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_host, d_out, n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_debug_eval: %s", hipGetErrorString(e));
    return RT_OK;
}

}  // extern "C"
