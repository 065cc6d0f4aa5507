// rt_api.hip -- the C ABI of include/rt_api.h: context = the reference's OpenCLConfigBuffer
// (SimpleRT/src/OpenCLConfig.cpp:398-747) re-done for one MI355X: device buffers, scene
// tables, launch geometry, row-tile sharding, counters.  No CPU fallback: without a HIP
// device every entry point fails with RT_ERR_NO_DEVICE.
//
// Ordering rule of this file: ALL device work of one context -- launches, resets, scene updates,
// read-backs -- executes in the order the calls were made, whatever streams the caller passes:
// when a call uses another stream than the context's previous piece of work, the library records an
// event on the previous stream and makes the new one wait for it (chain()).  Nothing here uses the
// null stream or a device-wide synchronisation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <mutex>
#include <new>
#include <vector>

#include "rt_internal.h"

#ifndef RT_DIAGNOSTICS
#define RT_DIAGNOSTICS 0
#endif

namespace {
thread_local char g_err[512] = "";
// where the last rt_create / rt_create_sharded spent its host time, in milliseconds (rt_debug_create_breakdown):
// [0] device query, [1] stream + events, [2] device allocations, [3] kernel function attributes (the first context of a
// process pays the load of the library's code object here), [4] default seed stream generated on the host, [5] its upload,
// [6] the restore kernel (first launch) and the wait for it, [7] total
double g_create_ms[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}
}

namespace rt {
int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace rt

using rt::fail;

// rt_reset_async zeroes the work counters with a kernel on the caller's stream and restores NO seeds:
// the next launch reads the pristine default stream directly (LaunchParams::seeds_in).
__global__ void rt_zero_counters_kernel(unsigned long long *counters, unsigned long long *stats) {
    for (int i = threadIdx.x; i < 32; i += blockDim.x) counters[i] = 0ull;
    for (int i = threadIdx.x; i < rt::kStatReplicas * 8; i += blockDim.x) stats[i] = 0ull;
}

// rt_create / rt_reset: seeds = the pristine default stream, colour plane, pixels and counters zero
// (OpenCLConfig.cpp:613-682), as ONE kernel instead of a runtime copy and four fills.
__global__ void __launch_bounds__(256) rt_restore_kernel(unsigned long long *seeds, const unsigned long long *seeds0, size_t n_pairs,
                                                         uint32_t *colors, size_t n_colors, uint32_t *pixels, size_t n_pixels,
                                                         unsigned long long *counters, unsigned long long *stats) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = first; i < n_pairs; i += stride) seeds[i] = seeds0[i];
    for (size_t i = first; i < n_colors; i += stride) colors[i] = 0u;
    for (size_t i = first; i < n_pixels; i += stride) pixels[i] = 0u;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < 32; i += blockDim.x) counters[i] = 0ull;
        for (int i = threadIdx.x; i < rt::kStatReplicas * 8; i += blockDim.x) stats[i] = 0ull;
    }
}

// Scene tables from the raw 44-byte records (rt_set_scene / rt_update_spheres_async), ONE workgroup:
//   geom[i]   = { p, rad*rad }               .cl:184      emis[i] = { e, bits(refl) }
//   colr[i]   = { c, rad }
//   lightA[j] = { p, rad }, lightB[j] = { e, 4*pi*rad*rad }   for the j-th sphere, in scene order, that
//   passes the reference's zero test (.cl:135-138: x and z only) -- the list SampleLights walks (.cl:249-303).
// Binary32, one operation per source operation (this file is compiled -ffp-contract=off): the same bits
// as the reference's `rad * rad` and `4.f * FLOAT_PI * rad * rad` (.cl:297) evaluated per use.
__global__ void __launch_bounds__(256) rt_build_tables_kernel(const rt_sphere *sph, uint32_t n, float4 *geom, float4 *emis,
                                                              float4 *colr, float4 *la, float4 *lb, uint32_t *n_lights_out) {
    __shared__ uint32_t s_wave_count[4];
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 256) {
        const uint32_t i = i0 + (uint32_t)tid;
        bool light = false;
        float rad = 0.f, px = 0.f, py = 0.f, pz = 0.f, ex = 0.f, ey = 0.f, ez = 0.f;
        if (i < n) {
            const float *r = reinterpret_cast<const float *>(sph + i);      // 11 dwords: rad, p, e, c, refl
            rad = r[0]; px = r[1]; py = r[2]; pz = r[3]; ex = r[4]; ey = r[5]; ez = r[6];
            geom[i] = make_float4(px, py, pz, rad * rad);
            emis[i] = make_float4(ex, ey, ez, r[10]);                        // refl keeps its bits
            colr[i] = make_float4(r[7], r[8], r[9], rad);
            light = !((ex == 0.f) && (ez == 0.f));
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(light);
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if ((tid & 63) == 0) s_wave_count[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t off = s_base;
        for (int k = 0; k < wave; ++k) off += s_wave_count[k];
        if (light) {
            la[off + before] = make_float4(px, py, pz, rad);
            lb[off + before] = make_float4(ex, ey, ez, 4.f * 3.14159265358979323846f * rad * rad);
        }
        __syncthreads();
        if (tid == 0) s_base += s_wave_count[0] + s_wave_count[1] + s_wave_count[2] + s_wave_count[3];
        __syncthreads();
    }
    if (tid == 0) *n_lights_out = s_base;
}

// Heavy-first order of the tiles from the costs the last launch left (rt_trace.inc.h): ONE workgroup; a counting sort over
// 1024 cost classes (largest first; the order inside a class does not matter).
// With n_home > 1 (rt_debug_set_tile_order; NOT the default: it saves a fifth of the launch's traffic and costs 1 % of its time)
// the order also keeps the tiles of a REGION (region_tx x region_ty tiles: the 32 x 32 pixels whose lanes the
// deal by cost mixes) on one XCD: workgroups are dealt to the 8 XCDs round-robin (block b and b + 8 share one: observed, not
// promised -- only traffic depends on it), so every region gets a home ((column + 3 x row) mod n_home), each home's tiles are sorted
// heavy first on their own, and position n_home * k + h takes the k-th tile of home h.  The wavefronts of a region then store
// their scattered pixels, colours and seeds through ONE L2, where the partial lines meet before they leave, and read the
// region's seeds and deal from it.  (Homes hold equally many tiles up to a region or two; the tiles beyond the shortest
// list's length -- the cheapest ones -- follow at the end.)
__global__ void __launch_bounds__(1024) rt_order_tiles_kernel(const uint32_t *cost, uint32_t *order, uint32_t n, uint32_t grid_x,
                                                             uint32_t region_tx, uint32_t region_ty, uint32_t n_home) {
    constexpr unsigned kMaxHome = 8;
    __shared__ unsigned s_max;
    __shared__ unsigned s_hist[kMaxHome][1024];
    __shared__ unsigned s_len[kMaxHome], s_tail[kMaxHome], s_min;
    const unsigned tid = threadIdx.x;
    constexpr unsigned kCap = 0x1FFFFFu;            // 21 ms of ticks: cost * 1023 stays inside 32 bits
    if (n_home < 1u || n_home > kMaxHome) n_home = 1u;
    auto home_of = [&](uint32_t i) -> unsigned {
        if (n_home == 1u) return 0u;
        const uint32_t ty = i / grid_x, tx = i - ty * grid_x;
        return (tx / region_tx + 3u * (ty / region_ty)) % n_home;       // (neighbours across AND down get different homes: a tall or a wide expensive object is spread over all of them)
    };
    auto key_of = [&](uint32_t i) -> unsigned { return cost[i] < kCap ? cost[i] : kCap; };
    if (tid == 0) s_max = 1u;
    for (unsigned h = 0; h < kMaxHome; ++h) s_hist[h][tid] = 0u;
    __syncthreads();
    unsigned m = 0;
    for (uint32_t i = tid; i < n; i += 1024) {
        const unsigned c_ = key_of(i);
        m = c_ > m ? c_ : m;
    }
    atomicMax(&s_max, m);
    __syncthreads();
    const unsigned top = s_max;
    for (uint32_t i = tid; i < n; i += 1024) {
        const unsigned c_ = key_of(i);
        atomicAdd(&s_hist[home_of(i)][1023u - c_ * 1023u / top], 1u);
    }
    __syncthreads();
    if (tid < kMaxHome) {               // exclusive prefix over the classes of one home, most expensive class first
        unsigned run = 0;
        for (int k = 0; k < 1024; ++k) {
            const unsigned c_ = s_hist[tid][k];
            s_hist[tid][k] = run;
            run += c_;
        }
        s_len[tid] = run;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned lo = 0xffffffffu;
        for (unsigned h = 0; h < n_home; ++h) lo = s_len[h] < lo ? s_len[h] : lo;
        s_min = lo;
        unsigned run = lo * n_home;
        for (unsigned h = 0; h < n_home; ++h) {
            s_tail[h] = run;
            run += s_len[h] - lo;
        }
    }
    __syncthreads();
    const unsigned shortest = s_min;
    for (uint32_t i = tid; i < n; i += 1024) {
        const unsigned c_ = key_of(i);
        const unsigned h = home_of(i);
        const unsigned k = atomicAdd(&s_hist[h][1023u - c_ * 1023u / top], 1u);
        order[k < shortest ? k * n_home + h : s_tail[h] + (k - shortest)] = i;
    }
}

// The deal of a region's pixels to its wavefronts (rt_device.h LaunchParams::deal): ONE workgroup per region of 32 x deal_rows
// pixels sorts them by the cost the last launch left for them (rays traced; descending, ties by position) -- a bitonic sort
// in LDS -- and writes their positions (dy * 32 + dx) in that order.  What is sorted are runs of `group` horizontally adjacent
// pixels (1, 2, 4 or 8; key = the run's summed cost; 8 by default): a run stays on adjacent lanes, so the launch's loads and
// stores of seeds, colours and pixels still come in segments of 8 * group .. 12 * group bytes instead of single words (with
// single pixels the launch wrote 3.5 times the bytes it produces).  What the previous launch cost predicts the next launch only
// as far as a pixel's EXPECTED cost goes -- single pixels sorted by the realised cost are an exact fit for the same frame rendered
// again (same random numbers) and a slight loss on new passes; runs of 4 and 8 gain on both (tools/deal_progressive.py).  A region that is not wholly inside the rendered rows keeps the 8x8 squares
// (its workgroups do not all exist: ranks must not move out of their square).
__global__ void __launch_bounds__(1024) rt_order_pixels_kernel(const uint16_t *__restrict__ cost, uint16_t *__restrict__ deal, int w, int rows,
                                                              int regions_x, int deal_rows, int group) {
    __shared__ uint32_t s_key[rt::kRegionW * rt::kMaxDealRows];
    const int tid = threadIdx.x, nt = blockDim.x, n = rt::kRegionW * deal_rows;       // n pixels: a power of two
    const int ng = n / group;                                                          // runs: a power of two as well
    const int region = blockIdx.x, ry = region / regions_x, rx = region - ry * regions_x;
    const int x0 = rx * rt::kRegionW, y0 = ry * deal_rows;
    const bool whole = (x0 + rt::kRegionW <= w) && (y0 + deal_rows <= rows);
    if (!whole) {
        // identity: rank (band b, wavefront q, lane l) -> the pixel (q * 8 + (l & 7), b * 8 + (l >> 3)) of the wavefront's own square
        for (int r = tid; r < n; r += nt) {
            const int b = r >> 8, q = (r >> 6) & 3, l = r & 63;
            deal[(size_t)region * n + r] = (uint16_t)(((b * 8 + (l >> 3)) << 5) | (q * 8 + (l & 7)));
        }
        return;
    }
    for (int i = tid; i < ng; i += nt) {
        const int p0 = i * group, dx = p0 & 31, dy = p0 >> 5;                          // (a run never crosses a row: 32 % group == 0)
        uint32_t c_ = 0;
        for (int j = 0; j < group; ++j) c_ += cost[(size_t)(y0 + dy) * (size_t)w + (size_t)(x0 + dx + j)];
        s_key[i] = (c_ << 12) | (uint32_t)(4095 - i);       // descending sort of the key = heaviest run first, then lowest position
    }
    __syncthreads();
    for (int k = 2; k <= ng; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < ng; i += nt) {
                const int l = i ^ j;
                if (l > i) {
                    const uint32_t a = s_key[i], b = s_key[l];
                    if ((a < b) == ((i & k) == 0)) { s_key[i] = b; s_key[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < ng; i += nt) {
        const uint32_t p0 = (4095u - (s_key[i] & 4095u)) * (uint32_t)group;
        for (int j = 0; j < group; ++j) deal[(size_t)region * n + (size_t)i * group + j] = (uint16_t)(p0 + (uint32_t)j);
    }
}

// rt_deinterleave_rows: full[y] = row (t/n)*tile_rows + y%tile_rows of rank t%n's block, t = y/tile_rows.
// One thread per 16 bytes where the row length allows it (w % 4 == 0 keeps every row 16-byte aligned).
__global__ void __launch_bounds__(256) rt_deinterleave_kernel(uint32_t *__restrict__ full, const uint32_t *__restrict__ gathered, int w,
                                                              int h, int nranks, int tile_rows, int pad_rows, int vec) {
    const int per_row = vec ? w / 4 : w;
    const size_t total = (size_t)per_row * (size_t)h;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / (size_t)per_row), xq = (int)(i - (size_t)y * (size_t)per_row);
        const int t = y / tile_rows, r = t % nranks, lrow = (t / nranks) * tile_rows + (y - t * tile_rows);
        const size_t src = ((size_t)r * (size_t)pad_rows + (size_t)lrow) * (size_t)w, dst = (size_t)y * (size_t)w;
        if (vec)
            reinterpret_cast<uint4 *>(full + dst)[xq] = reinterpret_cast<const uint4 *>(gathered + src)[xq];
        else
            full[dst + xq] = gathered[src + xq];
    }
}

#if RT_DIAGNOSTICS
// diagnostic only (rt_debug_stage_tables): EXACTLY the table staging of the render kernels' prologue -- every workgroup reads the
// geometry, light and (if they ride along) material tables into LDS -- and nothing else, so that the L2 counters of a profiler
// run show the hit rate of those reads in isolation (north_star: "L2-hit rate on the LDS-staged sphere reads").  One word per
// workgroup goes out so that the loads are not dead.
__global__ void rt_stage_probe_kernel(const rt::SceneTables T, int mat_in_lds, uint32_t *sink) {
    extern __shared__ float4 lds[];
    const uint32_t n = T.n_spheres, nl = T.n_lights;
    float4 *s_geom = lds, *s_la = s_geom + n, *s_lb = s_la + nl, *s_em = s_lb + nl, *s_co = s_em + n;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) s_geom[i] = T.geom[i];
    for (uint32_t i = threadIdx.x; i < nl; i += blockDim.x) {
        s_la[i] = T.lightA[i];
        s_lb[i] = T.lightB[i];
    }
    if (mat_in_lds)
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            s_em[i] = T.emis[i];
            s_co[i] = T.colr[i];
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        float acc = 0.f;
        for (uint32_t i = 0; i < n; ++i) acc += s_geom[i].w + (mat_in_lds ? s_em[i].x + s_co[i].x : 0.f);
        for (uint32_t i = 0; i < nl; ++i) acc += s_la[i].w + s_lb[i].w;
        sink[blockIdx.x & 1023u] = __float_as_uint(acc);
    }
}

// diagnostic only (rt_debug_reset_by_copy): the reset this library used in round 1 -- a copy kernel that
// restores the seed words, which the next launch then reads back.  Logs into its timelog record (tl) the
// device wall-clock of its first start / last end and the number of workgroups that ran, and per workgroup
// (blocklog) its start time and the XCD it ran on.  flags bit 1: every wave ends with an explicit
// agent-scope release (buffer_wbl2 sc1 + wait), i.e. the shader itself writes its XCD's L2 back instead
// of leaving that to the end-of-kernel action of the command processor.  flags bit 2: write-through stores; bit 3: atomic exchanges instead of stores.
__global__ void rt_debug_copy_seeds_kernel(uint32_t *seeds, const uint32_t *seeds0, size_t n, unsigned long long *tl,
                                           unsigned long long tag, unsigned long long *blocklog, int flags) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (tl && threadIdx.x == 0) atomicMin(&tl[0], t0);
    if (tl && threadIdx.x == 0 && blockIdx.x == 0) { tl[2] = 2ull; tl[3] = tag; }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (flags & 8) __hip_atomic_exchange(seeds + i, seeds0[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // memory-side read-modify-write
        else if (flags & 4) __hip_atomic_store(seeds + i, seeds0[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through (global_store ... sc1)
        else seeds[i] = seeds0[i];
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (flags & 2) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (threadIdx.x == 0) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (blocklog && (flags & 8)) __hip_atomic_exchange(&blocklog[blockIdx.x], (t0 << 4) | (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (blocklog) blocklog[blockIdx.x] = (t0 << 4) | (xcc & 15u);
        if (tl) atomicAdd(&tl[5], 1ull);                                    // workgroups that ran
        if (tl) atomicMax(&tl[1], __builtin_amdgcn_s_memrealtime());
    }
}

// flags bit 0: every wave starts with an explicit agent-scope acquire (buffer_inv sc1 + wait) before it reads
__global__ void rt_debug_probe_seeds_kernel(const uint32_t *seeds, const uint32_t *seeds0, size_t n, unsigned long long *out,
                                            unsigned long long *tl, unsigned long long tag, uint32_t *stalelog, int flags) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (tl && threadIdx.x == 0) atomicMin(&tl[0], t0);
    if (tl && threadIdx.x == 0 && blockIdx.x == 0) { tl[2] = 3ull; tl[3] = tag; }
    if (flags & 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (seeds[i] != seeds0[i]) {
            b += 1;
            if (stalelog) {                                                  // first 63 stale words: index | reader's XCD << 28
                const uint32_t k = atomicAdd(&stalelog[0], 1u);
                if (k < 63u) stalelog[1 + k] = (uint32_t)i | (xcc << 28);
            }
        }
    }
    if (b) atomicAdd(out, b);
    if (b && tl) atomicAdd(&tl[4], b);                                      // stale words seen by THIS probe
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(out + 1, 1ull);     // probes run
    if (tl && threadIdx.x == 0) atomicMax(&tl[1], __builtin_amdgcn_s_memrealtime());
}
#endif

namespace {

int select_device(const rt_ctx *c) {
    HIP_TRY(hipSetDevice(c->device));
    return RT_OK;
}

// the context's work runs in issue order: `stream` waits for whatever the context queued last elsewhere
int chain(rt_ctx *c, hipStream_t stream) {
    if (c->last_stream != stream) {
        HIP_TRY(hipEventRecord(c->ev_dep, c->last_stream));
        HIP_TRY(hipStreamWaitEvent(stream, c->ev_dep, 0));
        c->last_stream = stream;
        if (stream != c->stream) c->used_foreign_stream = true;
    }
    return RT_OK;
}

// host waits for everything the context has queued
int wait_all(rt_ctx *c) {
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    return RT_OK;
}

int upload_default_seeds(rt_ctx *c) {
    const size_t count = 2 * (size_t)c->w * (size_t)c->h;
    const double t0 = now_ms();
    std::vector<uint32_t> host(count);
    rt_default_seeds(host.data(), count);
    g_create_ms[4] = now_ms() - t0;
    HIP_TRY(hipMemcpyAsync(c->d_seeds0, host.data(), count * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));          // `host` goes out of scope
    return RT_OK;
}

int restore_state(rt_ctx *c) {                          // rt_create / rt_reset: blocking
    int rc = chain(c, c->stream);
    if (rc != RT_OK) return rc;
    const size_t px = (size_t)c->w * (size_t)c->h;
    size_t blocks = (3 * px + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rt_restore_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, reinterpret_cast<unsigned long long *>(c->d_seeds),
                       reinterpret_cast<const unsigned long long *>(c->d_seeds0), px, reinterpret_cast<uint32_t *>(c->d_colors), 3 * px,
                       c->d_pixels, (size_t)c->local_rows * (size_t)c->w, c->d_counters, c->d_stats);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

rt::LaunchParams make_params(rt_ctx *c, int n_samples) {
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
    p.coop_kmax = c->coop_kmax;
    p.tiles_x = (c->w + 7) / 8;
    p.n_tiles = p.tiles_x * ((c->local_rows + 7) / 8);
    return p;
}

constexpr size_t kLdsMax = 152 * 1024;     // what the kernels' dynamic-LDS attribute allows

// ---- which instance, and what it needs (rt_device.h Instance; the rows live next to the instantiations) ----

const rt::Instance *instances(bool fast, int *count) { return fast ? rt::fast_instances(count) : rt::parity_instances(count); }

// the row with this role and workgroup shape (null: this library has none)
const rt::Instance *find_role(bool fast, int role, int waves) {
    int n = 0;
    const rt::Instance *t = instances(fast, &n);
    for (int k = 0; k < n; ++k)
        if (t[k].role == role && t[k].waves == waves) return &t[k];
    return nullptr;
}

// LDS the hierarchy's staged tables take for this scene
size_t pairs_lds(const rt_ctx *c, bool mat, int n_samples) {
    return rt::lds_bytes_pairs(c->scene.n_spheres, c->scene.n_lights, mat, n_samples, c->bvh.n_leaves, c->bvh.n_slots, c->bvh.stack_depth, 256);
}

// the plain sweep's tables (geometry and lights) fit LDS for this launch
bool tables_fit_lds(const rt_ctx *c, int n_samples) {
    return rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, false, n_samples) <= kLdsMax;
}

// the hierarchy's tables fit the LDS budget given to them; otherwise the walk reads them from HBM / L2
bool bvh_fits_lds(const rt_ctx *c, int n_samples) { return pairs_lds(c, false, n_samples) <= (size_t)c->bvh_lds_limit; }

// the scene has a hierarchy and the context may use it
bool bvh_usable(const rt_ctx *c) { return c->bvh_ok && c->wg_waves != 1 && c->persist == 0; }

// What the instance needs from the context, checked against what the context has: the ONE place that sizes the
// dynamic LDS and hands out the hierarchy.  An instance whose tables the context lacks is refused (RT_ERR_STATE),
// whatever route selected it -- the measured choice, a forced form, or a diagnostics mode.
int bind_tables(rt_ctx *c, const rt::Instance &inst, int n_samples, rt::LaunchParams &p, size_t *lds_out) {
    const bool needs_bvh = inst.tables == rt::kTabPairsLds || inst.tables == rt::kTabPairsGlobal;
    p.bvh = rt::BvhTables{};
    if (needs_bvh) {
        if (!c->bvh_ok || !c->bvh.blob)
            return fail(RT_ERR_STATE, "%s walks a hierarchy and the scene has none (fewer than %d small spheres?)", inst.name, c->bvh_min);
        p.bvh = c->bvh;
    }
    size_t lds = 0;
    switch (inst.tables) {
        case rt::kTabSweepLds:
            lds = rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, p.mat_in_lds != 0, n_samples);
            break;
        case rt::kTabSweepGlobal:
            p.mat_in_lds = 0;
            lds = rt::lds_bytes(0, 0, false, n_samples);
            break;
        case rt::kTabPairsLds:
            p.mat_in_lds = 0;               // (the walk reads a hit's material by slot from the hierarchy's blob: nothing of it is staged)
            lds = pairs_lds(c, false, n_samples);
            break;
        case rt::kTabPairsGlobal:
            p.mat_in_lds = 0;
            lds = rt::lds_bytes_pairs(0, 0, false, n_samples, 1, 0, c->bvh.stack_depth, 256);
            break;
        default:
            return fail(RT_ERR_STATE, "%s: unknown table kind %d", inst.name, inst.tables);
    }
    if (lds > kLdsMax) return fail(RT_ERR_ARG, "%s needs %zu B of LDS for this scene (limit %zu)", inst.name, lds, kLdsMax);
    *lds_out = lds;
    return RT_OK;
}

// `form`: 0 = the context's choice, 1 = the hierarchy (if the scene has one), 2 = the plain sweep
int launch_form(rt_ctx *c, int n_samples, hipStream_t stream, int form, bool natural_order = false) {
    if (!c->have_scene || !c->have_cam)
        return fail(RT_ERR_STATE, "rt_set_scene and rt_set_camera must precede rendering");
    if (n_samples < 0) return fail(RT_ERR_ARG, "n_samples < 0");
    if (n_samples > 0x7fffffff - c->current_sample)
        return fail(RT_ERR_ARG, "pass counter would overflow (%d + %d)", c->current_sample, n_samples);
    if (n_samples == 0 || c->local_rows == 0) return RT_OK;
    int rc = chain(c, stream);
    if (rc != RT_OK) return rc;

    rt::LaunchParams p = make_params(c, n_samples);
    const size_t lds_all = rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, true, n_samples);
    // materials ride along in LDS only while that keeps at least 6 workgroups per CU resident
    // (160 KiB / 24 KiB); larger scenes read them from L2 once per hit
    p.mat_in_lds = lds_all <= (size_t)c->mat_lds_limit;
    const size_t lds_sweep = rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, p.mat_in_lds != 0, n_samples);

    // which instance: arithmetic mode x role x workgroup shape.  Single-wavefront workgroups (8x8 tiles) keep the wave
    // slots of a CU full (a 4-wavefront workgroup waits for four free slots at once) and give the heavy-first order a
    // finer granule; each stages its own copy of the tables, so only while 24 copies fit a CU.
    bool fast = c->mode == RT_MODE_FAST;
    const bool coop = c->coop_min > 0 && c->scene.n_spheres >= (uint32_t)c->coop_min;
    const bool w1 = c->wg_waves == 1 || (c->wg_waves == 0 && lds_sweep + (coop ? 1536u : 256u) <= 6 * 1024);   // + the instance's static LDS
    int role = coop ? rt::kRoleCoop : rt::kRolePlain, waves = w1 ? 1 : 4;
    if (form != 2 && bvh_usable(c)) {
        // large scenes: the walk over the hierarchy, from LDS while its tables leave room for five workgroups per CU
        role = bvh_fits_lds(c, n_samples) ? rt::kRolePairs : rt::kRolePairsGlobal;
        waves = 4;
        if (c->regen_gate <= 0) p.regen_gate = c->walk_gate;
    } else if (!tables_fit_lds(c, n_samples)) {
        // no hierarchy (or it lost the measurement) and a table beyond LDS: the plain sweep over the table in HBM / L2
        role = rt::kRoleSweepGlobal;
        waves = 4;
    }
    p.walk_round = c->walk_round;
    const rt::Instance *inst = nullptr;
#if RT_DIAGNOSTICS
    if (c->persist != 0 && c->mode < 100) {
        role = coop ? rt::kRolePersistCoop : rt::kRolePersist;
        waves = 4;
    } else if (c->mode >= 100) {           // a row of the table by number (rt_set_mode checked the range)
        fast = c->mode >= 200;
        int n = 0;
        const rt::Instance *t = instances(fast, &n);
        inst = &t[c->mode - (fast ? 200 : 100)];
    }
#endif
    if (!inst) inst = find_role(fast, role, waves);
    if (!inst) return fail(RT_ERR_STATE, "this library holds no %s instance of role %d with %d wavefronts per workgroup", fast ? "fast" : "parity", role, waves);
    size_t lds_use = 0;
    rc = bind_tables(c, *inst, n_samples, p, &lds_use);
    if (rc != RT_OK) return rc;
    const bool persist = (inst->flags & rt::kInstPersistent) != 0;

    const int tile_w = 8 * inst->waves;
    dim3 grid((unsigned)((c->w + tile_w - 1) / tile_w), (unsigned)((c->local_rows + rt::kTileH - 1) / rt::kTileH));
    // heavy tiles first: every launch leaves per-tile costs; once a long launch has, the next long launch of the
    // same scene, camera and tile shape walks the tiles in descending order of cost (sorted on the device, once)
    const uint32_t n_tiles = grid.x * grid.y;
    const bool instance_logs_cost = (inst->flags & rt::kInstNoTileCost) == 0;
    if (c->use_order && c->d_tile_cost && n_tiles <= c->n_tiles && instance_logs_cost) {
        // pixels dealt to wavefronts by cost: every launch leaves the rays it traced per pixel; once a long launch has,
        // the 256 pixels of each 32x8 region are sorted by them (on the device, once per scene and camera) and later
        // launches hand rank r of a region to wavefront r / 64, lane r % 64.  The tile costs measured under the old
        // deal no longer describe the workgroups: the heavy-first order is sorted again from the next launch's.
        if (c->use_deal && c->d_pixel_cost && !persist) {
            // only launches of 8 passes and more leave per-pixel costs (fewer are mostly noise, and the pricing launches and the
            // adapter's small batches would overwrite a good plane with them); the unit is the form's own -- loop trips of the
            // sweep kernels, rays of the walk -- so costs written by the other form are not sorted from
            const int form_now = (inst->tables == rt::kTabPairsLds || inst->tables == rt::kTabPairsGlobal) ? 1 : 2;
            if (c->pixel_cost_valid && c->pixel_cost_form != form_now) c->pixel_cost_valid = false;
            if (n_samples >= 8) p.pixel_cost = c->d_pixel_cost;
            if (c->pixel_cost_valid && !c->deal_valid && n_samples >= 4 && !natural_order) {
                const int regions_x = (c->w + rt::kRegionW - 1) / rt::kRegionW, regions_y = (c->local_rows + c->deal_rows - 1) / c->deal_rows;
                hipLaunchKernelGGL(rt_order_pixels_kernel, dim3((unsigned)(regions_x * regions_y)), dim3((unsigned)std::min(rt::kRegionW * c->deal_rows, 1024)), 0, stream,
                                   c->d_pixel_cost, c->d_deal, c->w, c->local_rows, regions_x, c->deal_rows, c->deal_group);
                HIP_TRY(hipGetLastError());
                c->deal_valid = true;
                c->cost_valid = c->order_valid = false;
            }
            if (c->deal_valid) {
                p.deal = c->d_deal;
                p.deal_rows = c->deal_rows;
            }
        }
        if (c->cost_tiles != n_tiles) c->cost_valid = c->order_valid = false;      // another tile shape: start over
        p.tile_cost = c->d_tile_cost;
        if (c->cost_valid && !c->order_valid && n_samples >= 8 && !natural_order) {
            // (a region: 32 pixels across = 4 single-wavefront tiles or one 4-wavefront tile; the deal's rows down)
            hipLaunchKernelGGL(rt_order_tiles_kernel, dim3(1), dim3(1024), 0, stream, c->d_tile_cost, c->d_order, n_tiles, grid.x,
                               inst->waves == 1 ? 4u : 1u, (uint32_t)(c->deal_rows / rt::kTileH), (uint32_t)c->order_homes);
            HIP_TRY(hipGetLastError());
            c->order_valid = true;
            c->order_age = 0;
        }
        if (c->order_valid && !natural_order) p.order = c->d_order;
    }
    if (persist) {
        // just enough workgroups to fill the machine; the tile queue (counters[30]) does the rest
        size_t per_cu = lds_use > 0 ? (160 * 1024) / (lds_use + 6 * 1024) : 6;
        if (per_cu > 6) per_cu = 6;
        if (per_cu < 1) per_cu = 1;
        size_t blocks = (size_t)c->n_cus * per_cu;
        const size_t needed = ((size_t)p.n_tiles + 3) / 4;
        if (blocks > needed) blocks = needed;
        grid = dim3((unsigned)blocks, 1, 1);
        HIP_TRY(hipMemsetAsync(c->d_counters + 30, 0, sizeof(unsigned long long), stream));
    }
#if RT_DIAGNOSTICS
    if (inst->role == rt::kRoleTimelog && c->d_timelog && c->timelog_used < c->timelog_cap) {
        p.timelog = c->d_timelog;
        p.seq = c->timelog_used++;
        p.tl_tag = c->timelog_tag;
        p.wavelog = ((size_t)grid.x * grid.y * 4 <= c->wavelog_cap) ? c->d_wavelog : nullptr;
    }
#endif
    const hipError_t e = rt::launch_instance(*inst, p, grid, lds_use, stream);
    if (e != hipSuccess)
        return fail(RT_ERR_HIP, "kernel launch failed: %s (%s, grid %ux%u, lds %zu B)", hipGetErrorString(e), inst->name, grid.x, grid.y, lds_use);
    c->current_sample += n_samples;
    c->launches += 1;
    c->last_kernel = inst->name;
    c->last_form = (inst->tables == rt::kTabPairsLds || inst->tables == rt::kTabPairsGlobal) ? 1 : 2;
    if (p.tile_cost && n_samples >= 4) {
        c->cost_valid = true;
        c->cost_tiles = n_tiles;
    }
    if (p.pixel_cost) {
        c->pixel_cost_valid = true;
        c->pixel_cost_form = c->last_form;
    }
    c->seeds_default = false;           // this launch has written every seed pair the context renders
    c->pixels_current = c->pixel_write != 0;
    return RT_OK;
}

// Hierarchy or plain sweep for this scene?  The walk wins by 5x on a thousand spheres scattered over a plane and
// loses on a box packed with overlapping glass -- so it is measured, once per scene: four launches in the same
// (natural) tile order -- the hierarchy warm, the hierarchy timed, the sweep warm, the sweep timed, each timed one
// between two events -- and when both timings have arrived (asked without blocking) the form that took less time per
// pass renders the rest.  A blocking call with enough passes splits off 1 + 2 + 1 + 2 passes for the probes and waits
// for the verdict before it queues the rest (progressive passes equal one launch bit for bit).  The verdict is kept
// for the scene; device-resident updates keep it until the tree has changed size by a quarter or 256 updates have
// gone by (rearm_probe_if_changed).  In a multi-device context only the first shard measures; the others follow it.
constexpr int kProbeSteps = 4;          // hierarchy warm, hierarchy timed, sweep warm, sweep timed

void probe_poll(rt_ctx *c, bool wait) {
    if (c->bvh_pick != 0 || c->probe_state < kProbeSteps) return;
    if (wait) {
        if (hipEventSynchronize(c->probe_ev[3]) != hipSuccess) return;
    } else if (hipEventQuery(c->probe_ev[3]) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    float a = 0.f, b = 0.f;
    if (hipEventElapsedTime(&a, c->probe_ev[0], c->probe_ev[1]) != hipSuccess || hipEventElapsedTime(&b, c->probe_ev[2], c->probe_ev[3]) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    const double ta = (double)a / c->probe_samples[0], tb = (double)b / c->probe_samples[1];
    c->probe_ms[0] = ta;
    c->probe_ms[1] = tb;
    c->bvh_pick = ta <= 1.05 * tb ? 1 : 2;      // (a dead band of 5 % towards the usual winner: no flipping on a tie)
    c->pick_estimated = false;
    c->probe_tree = c->bvh_n_tree;
    c->probe_always = c->bvh.n_always;
    c->probe_updates = 0;
}

int launch_probe(rt_ctx *c, int n_samples, hipStream_t stream) {
    const int k = c->probe_state;               // 0, 1: hierarchy (warm, timed); 2, 3: plain sweep (warm, timed)
    const bool timed = (k & 1) != 0;
    const int arm = k >> 1;
    int rc = chain(c, stream);
    if (rc != RT_OK) return rc;
    if (timed) HIP_TRY(hipEventRecord(c->probe_ev[2 * arm], stream));
    rc = launch_form(c, n_samples, stream, arm == 0 ? 1 : 2, true);
    if (rc != RT_OK) return rc;
    if (timed) {
        HIP_TRY(hipEventRecord(c->probe_ev[2 * arm + 1], stream));
        c->probe_samples[arm] = n_samples;
    }
    c->probe_state = k + 1;
    return RT_OK;
}

void rearm_probe(rt_ctx *c) {
    c->bvh_pick = 0;
    c->pick_estimated = false;
    c->probe_state = 0;
    c->probe_ms[0] = c->probe_ms[1] = 0.0;
    c->probe_updates = 0;
}

// after a device-resident update rebuilt the hierarchy: is the verdict still about this tree?
void rearm_probe_if_changed(rt_ctx *c) {
    if (c->bvh_pick == 0 && c->probe_state == 0) return;
    if (!c->bvh_ok) {
        rearm_probe(c);
        return;
    }
    const uint32_t tree = c->bvh_n_tree, always = c->bvh.n_always;        // spheres, not padded slots: the shaped tree of an upload has partial leaves
    auto moved = [](uint32_t now, uint32_t then) { return 4u * (now > then ? now - then : then - now) > then + 8u; };
    if (moved(tree, c->probe_tree) || moved(always, c->probe_always) || ++c->probe_updates >= 256) {
        rearm_probe(c);
        c->bvh_est_valid = false;               // (the areas were the uploaded tree's: the changed scene is measured)
    }
}

// A long launch that would walk its tiles in image order although their costs can be had -- the first frame of a scene or camera,
// and the frame after it, whose deal of pixels changes what a tile is -- renders 4 of its passes first (they are passes of the
// frame like any other: progressive launches equal one launch bit for bit), which prices the tiles, and the rest heavy first.
// A renderer that draws one frame per scene would otherwise never leave image order ("Heavy tiles first").
constexpr int kPricePasses = 4, kPriceFrom = 24;
int launch_priced(rt_ctx *c, int n_samples, hipStream_t stream, int form) {
    const bool explicit_mode = form == 0;
    const bool will_deal = c->use_deal && c->pixel_cost_valid && !c->deal_valid;
    if (!explicit_mode && c->use_order && c->d_tile_cost && n_samples >= kPriceFrom && ((!c->order_valid && !c->cost_valid) || will_deal)) {
        const int rc = launch_form(c, kPricePasses, stream, form);
        if (rc != RT_OK) return rc;
        n_samples -= kPricePasses;
    }
    return launch_form(c, n_samples, stream, form);
}

// The same question answered WITHOUT a launch, from the surface areas of the tree the host built at rt_set_scene (rt_bvh.hip):
// a random line through the root box is expected to visit  P = sum of area(inner node) / area(root)  pairs (and leaves in
// proportion), each ray sweeps the always-list besides, and the plain sweep tests all n spheres.  Predicted time per ray of
// the walk over that of the sweep, in units of one sphere test of the sweep:
//     ratio = (kEstPair * P + kEstAlways * n_always) / (n + kEstSweepFixed)
// The three weights are a least-squares fit (log ratio) to the probe's own timings of both forms on 32 scenes of four
// families -- spheres scattered on a plane, a closed box packed with mirror / glass spheres, a cloud in the air, the Demo
// scene plus scattered spheres; 64 to 1400 spheres -- tools/choice_calibration.py, profiles/r04a_choice_calibration.jsonl:
// rms error 10 %, 8 % at worst between 0.6 and 1.7.  (A term for the expected leaf visits fitted to zero: they go with P.)
// Outside a band around 1 the estimate decides and nothing is measured -- a new scene's first frame then costs what a frame
// costs; inside it the four probe launches run as before.
constexpr double kEstPair = 20.9, kEstAlways = 10.1, kEstSweepFixed = 20.1;
constexpr double kEstBandLo = 0.75, kEstBandHi = 1.33;
double estimate_ratio(const rt_ctx *c) {
    return (kEstPair * c->bvh_est_pairs + kEstAlways * (double)c->bvh.n_always) / ((double)c->scene.n_spheres + kEstSweepFixed);
}

int launch(rt_ctx *c, int n_samples, hipStream_t stream, bool may_block = false) {
    const bool measured = c->walk_forced == 0 && c->mode < 100;
    if (!measured || n_samples <= 0 || c->local_rows == 0 || !c->have_scene || !c->have_cam || !bvh_usable(c))
        return measured ? launch_priced(c, n_samples, stream, 2) : launch_form(c, n_samples, stream, 0);
    // no probe where the answer is known and asking is dear: from 1500 spheres in the tree on the hierarchy won on every
    // scene measured, open or packed, and one pass of the sweep at 1080p costs 2.6 ms at 1024 spheres,
    // 28 ms at 4096, 138 ms at 8192, seconds beyond LDS
    if (c->bvh_n_tree >= 1500u || !tables_fit_lds(c, n_samples)) return launch_priced(c, n_samples, stream, 1);
    if (c->choice_leader)                       // a shard of a multi-device context: the form the first shard just launched
        return launch_priced(c, n_samples, stream, c->choice_leader->last_form == 2 ? 2 : 1);
    probe_poll(c, false);
    if (c->bvh_pick == 0 && c->probe_state == 0 && c->use_estimate && c->bvh_est_valid) {
        const double r = estimate_ratio(c);
        c->est_ratio = r;
        if (r < kEstBandLo || r > kEstBandHi) {
            c->bvh_pick = r < 1.0 ? 1 : 2;
            c->pick_estimated = true;
            c->probe_tree = c->bvh_n_tree;
            c->probe_always = c->bvh.n_always;
            c->probe_updates = 0;
        }
    }
    if (c->bvh_pick != 0) return launch_priced(c, n_samples, stream, c->bvh_pick);
    if (c->probe_state == kProbeSteps) return launch_form(c, n_samples, stream, 1);    // probes in flight: the usual winner meanwhile
    if (may_block && n_samples >= 16) {
        int done = 0;
        while (c->probe_state < kProbeSteps) {
            const int k = (c->probe_state & 1) ? 2 : 1;
            const int rc = launch_probe(c, k, stream);
            if (rc != RT_OK) return rc;
            done += k;
        }
        probe_poll(c, true);
        return launch_priced(c, n_samples - done, stream, c->bvh_pick ? c->bvh_pick : 1);
    }
    return launch_probe(c, n_samples, stream);
}

// ---- scene storage -------------------------------------------------------------------------

bool light_test(const rt_sphere &s) { return !((s.e.x == 0.f) && (s.e.z == 0.f)); }   // .cl:135-138,266

void free_scene(rt_ctx *c) {
    (void)hipFree(c->d_spheres);
    (void)hipFree(c->d_tables);
    (void)hipFree(c->d_bvh);
    c->d_spheres = nullptr;
    c->d_tables = nullptr;
    c->d_bvh = nullptr;
    c->bvh_ok = false;
    c->scene_cap = 0;
}

int ensure_scene_capacity(rt_ctx *c, uint32_t count) {
    if (count <= c->scene_cap && c->d_tables) return RT_OK;
    uint32_t cap = 64;
    while (cap < count) cap *= 2;
    int rc = wait_all(c);               // nothing may still read the tables that are about to go
    if (rc != RT_OK) return rc;
    rt_sphere *ns = nullptr;
    float4 *nt = nullptr;
    HIP_TRY(hipMalloc(&ns, (size_t)cap * sizeof(rt_sphere)));
    float4 *nb = nullptr;
    hipError_t e = hipMalloc(&nt, ((size_t)cap * 5 + 1) * sizeof(float4));
    // blob: hdr 2 + slots (< cap + 8; up to twice that with the partial leaves of the shaped tree) + index (a quarter of the
    // slots) + pairs (< cap / 2 + 4) + two material records per slot, in float4
    if (e == hipSuccess) e = hipMalloc(&nb, ((size_t)cap * 6 + 64) * sizeof(float4));
    if (e != hipSuccess) {
        (void)hipFree(ns);
        (void)hipFree(nt);
        return fail(RT_ERR_ALLOC, "scene tables for %u spheres: %s", cap, hipGetErrorString(e));
    }
    free_scene(c);
    c->d_spheres = ns;
    c->d_tables = nt;
    c->d_bvh = nb;
    c->scene_cap = cap;
    return RT_OK;
}

int ensure_stage_capacity(rt_ctx *c, uint32_t count) {
    if (count <= c->stage_cap) return RT_OK;
    uint32_t cap = 64;
    while (cap < count) cap *= 2;
    for (int k = 0; k < 4; ++k)
        if (c->stage_used[k]) {
            HIP_TRY(hipEventSynchronize(c->stage_ev[k]));
            c->stage_used[k] = false;
        }
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr;
    c->stage_cap = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_stage), (size_t)cap * 4 * sizeof(rt_sphere), hipHostMallocDefault));
    c->stage_cap = cap;
    return RT_OK;
}

// records [first, first+count) -> device, then the tables, all on `stream`.  `full_upload` is said by the caller, never inferred
// from the range: only rt_set_scene (which blocks anyway) may take the host-side build of the hierarchy; an update --
// whatever range it rewrites -- stays on the stream (rt_api.h: rt_update_spheres_async waits for nothing).
int upload_spheres(rt_ctx *c, uint32_t first, uint32_t count, const rt_sphere *spheres, uint32_t n_total, hipStream_t stream, bool full_upload) {
    int rc = chain(c, stream);
    if (rc != RT_OK) return rc;
    if (count) {
        rc = ensure_stage_capacity(c, count);
        if (rc != RT_OK) return rc;
        const int slot = c->stage_next;
        c->stage_next = (slot + 1) & 3;
        if (c->stage_used[slot]) HIP_TRY(hipEventSynchronize(c->stage_ev[slot]));
        rt_sphere *stage = c->h_stage + (size_t)slot * c->stage_cap;
        memcpy(stage, spheres, (size_t)count * sizeof(rt_sphere));
        HIP_TRY(hipMemcpyAsync(c->d_spheres + first, stage, (size_t)count * sizeof(rt_sphere), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(c->stage_ev[slot], stream));
        c->stage_used[slot] = true;
        for (uint32_t i = 0; i < count; ++i) c->is_light[first + i] = light_test(spheres[i]) ? 1 : 0;
    }
    uint32_t nl = 0;
    for (uint32_t i = 0; i < n_total; ++i) nl += c->is_light[i];
    const size_t cap = c->scene_cap;
    float4 *base = c->d_tables;
    float4 *d_geom = base, *d_emis = base + cap, *d_colr = base + 2 * cap, *d_la = base + 3 * cap, *d_lb = base + 4 * cap;
    if (n_total) {
        hipLaunchKernelGGL(rt_build_tables_kernel, dim3(1), dim3(256), 0, stream, c->d_spheres, n_total, d_geom, d_emis, d_colr,
                           d_la, d_lb, reinterpret_cast<uint32_t *>(base + 5 * cap));
        HIP_TRY(hipGetLastError());
    }
    c->scene = rt::SceneTables{ d_geom, d_emis, d_colr, d_la, d_lb, n_total, nl };
    return rt::build_bvh(c, n_total, stream, full_upload);
}

}  // namespace

namespace rt {
// a shard's launch on its own stream, for the multi-device context: like rt_render_async, but a BLOCKING frame (may_block) lets
// the shard that measures hierarchy against sweep hold all its probes inside this call, as rt_render_pass does
int render_shard(rt_ctx *c, int n_samples, bool may_block) {
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    return launch(c, n_samples, c->stream, may_block);
}
}  // namespace rt

extern "C" {

RT_API const char *rt_last_error(void) { return g_err; }

// used by rt_host.cpp (hidden: not part of the export table)
void rt_host_set_error(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg ? msg : ""); }

RT_API int rt_create_sharded(rt_ctx **out, int w, int h, int device, int rank, int nranks, int tile_rows) {
    if (!out) return fail(RT_ERR_ARG, "out is null");
    *out = nullptr;
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535) return fail(RT_ERR_ARG, "image size %dx%d (1 .. 65535 in either direction)", w, h);
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(RT_ERR_ARG, "rank %d of %d", rank, nranks);
    if (tile_rows <= 0 || tile_rows % rt::kTileH != 0)
        return fail(RT_ERR_ARG, "tile_rows must be a positive multiple of %d", rt::kTileH);

    const double t_begin = now_ms();
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
    g_create_ms[0] = now_ms() - t_begin;
    auto alloc_all = [&]() -> int {
        double t = now_ms();
        auto lap = [&](int k) { const double n_ = now_ms(); g_create_ms[k] = n_ - t; t = n_; };
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->last_stream = c->stream;
        HIP_TRY(hipEventCreate(&c->ev0));
        HIP_TRY(hipEventCreate(&c->ev1));
        for (int k = 0; k < 4; ++k) HIP_TRY(hipEventCreate(&c->probe_ev[k]));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_dep, hipEventDisableTiming));
        for (int k = 0; k < 4; ++k) HIP_TRY(hipEventCreateWithFlags(&c->stage_ev[k], hipEventDisableTiming));
        lap(1);
        HIP_TRY(hipMalloc(&c->d_seeds, 2 * px * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->d_seeds0, 2 * px * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->d_colors, 3 * px * sizeof(float)));
        HIP_TRY(hipMalloc(&c->d_pixels, ((size_t)rows * w + 4) * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->d_counters, 32 * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc(&c->d_stats, rt::kStatReplicas * 8 * sizeof(unsigned long long)));
        c->n_tiles = (uint32_t)(((w + 7) / 8) * ((rows + rt::kTileH - 1) / rt::kTileH));      // the finest tile shape (8x8)
        if (c->n_tiles) {
            HIP_TRY(hipMalloc(&c->d_tile_cost, (size_t)c->n_tiles * sizeof(uint32_t)));
            HIP_TRY(hipMalloc(&c->d_order, (size_t)c->n_tiles * sizeof(uint32_t)));
            // (regions of 8 rows need the most entries: every region is padded to whole rows of 32 pixels)
            const size_t deal_entries = (size_t)((w + rt::kRegionW - 1) / rt::kRegionW) * rt::kRegionW * (size_t)(((rows + 7) / 8) * 8 + rt::kMaxDealRows);
            HIP_TRY(hipMalloc(&c->d_pixel_cost, ((size_t)rows * w + 4) * sizeof(uint16_t)));
            HIP_TRY(hipMalloc(&c->d_deal, deal_entries * sizeof(uint16_t)));
        }
        lap(2);
        // function attributes (dynamic-LDS limit) are per device, not per context
        static std::mutex mu;
        static bool prepared[64] = {};
        {
            std::lock_guard<std::mutex> lock(mu);
            if (device >= 64 || !prepared[device]) {
                HIP_TRY(rt::prepare_parity());
                HIP_TRY(rt::prepare_fast());
                HIP_TRY(rt::prepare_bvh_build());
                if (device < 64) prepared[device] = true;
            }
        }
        lap(3);
        int r2 = upload_default_seeds(c);
        lap(5);
        g_create_ms[5] -= g_create_ms[4];
        if (r2 == RT_OK) r2 = restore_state(c);
        lap(6);
        g_create_ms[7] = now_ms() - t_begin;
        return r2;
    };
    if (rc == RT_OK) rc = alloc_all();
    if (rc != RT_OK) {
        rt_destroy(c);
        return rc;
    }
    *out = c;
    return RT_OK;
}

RT_API int rt_create(rt_ctx **out, int w, int h) { return rt_create_sharded(out, w, h, 0, 0, 1, rt::kTileH); }

RT_API void rt_destroy(rt_ctx *c) {
    if (!c) return;
    if (c->multi) {
        rt::multi_destroy(c);
        delete c;
        return;
    }
    if (hipSetDevice(c->device) == hipSuccess) {
        if (!c->abandon_streams) {
            if (c->last_stream && c->last_stream != c->stream) (void)hipStreamSynchronize(c->last_stream);
            if (c->stream) (void)hipStreamSynchronize(c->stream);
        }
        if (c->pinned_out) (void)hipHostUnregister(c->pinned_out);
        (void)hipFree(c->d_seeds);
        (void)hipFree(c->d_seeds0);
        (void)hipFree(c->d_colors);
        (void)hipFree(c->d_pixels);
        (void)hipFree(c->d_counters);
        (void)hipFree(c->d_stats);
        (void)hipFree(c->d_tile_cost);
        (void)hipFree(c->d_order);
        (void)hipFree(c->d_pixel_cost);
        (void)hipFree(c->d_deal);
        (void)hipFree(c->d_timelog);
        (void)hipFree(c->d_wavelog);
        (void)hipFree(c->d_blocklog);
        (void)hipFree(c->d_stalelog);
        free_scene(c);
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        if (c->ev0) (void)hipEventDestroy(c->ev0);
        if (c->ev1) (void)hipEventDestroy(c->ev1);
        for (int k = 0; k < 4; ++k)
            if (c->probe_ev[k]) (void)hipEventDestroy(c->probe_ev[k]);
        if (c->bvh_stage_ev) (void)hipEventDestroy(c->bvh_stage_ev);
        if (c->h_bvh_stage) (void)hipHostFree(c->h_bvh_stage);
        if (c->ev_dep) (void)hipEventDestroy(c->ev_dep);
        for (int k = 0; k < 4; ++k)
            if (c->stage_ev[k]) (void)hipEventDestroy(c->stage_ev[k]);
        for (auto &f : c->flight) {
            if (f.start) (void)hipEventDestroy(f.start);
            if (f.stop) (void)hipEventDestroy(f.stop);
        }
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

RT_API int rt_scene_choice(rt_ctx *c, double *hierarchy_ms_per_pass, double *sweep_ms_per_pass) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    rt_ctx *s = c->multi ? rt::multi_first_shard(c) : c;
    if (select_device(s) == RT_OK) probe_poll(s, false);
    if (hierarchy_ms_per_pass) *hierarchy_ms_per_pass = s->probe_ms[0];
    if (sweep_ms_per_pass) *sweep_ms_per_pass = s->probe_ms[1];
    return s->bvh_pick;
}

RT_API const char *rt_last_kernel(const rt_ctx *c) { return !c ? "" : (c->multi ? rt::multi_last_kernel(c) : c->last_kernel); }

RT_API int rt_shard_count(const rt_ctx *c) { return !c ? RT_ERR_ARG : (c->multi ? rt::multi_shards(c) : 1); }

RT_API int rt_set_scene(rt_ctx *c, const rt_sphere *spheres, uint32_t count) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    if (count > 0 && !spheres) return fail(RT_ERR_ARG, "spheres is null");
    if (count > RT_MAX_SPHERES) return fail(RT_ERR_ARG, "%u spheres > RT_MAX_SPHERES (%u)", count, RT_MAX_SPHERES);
    if (c->multi) return rt::multi_set_scene(c, spheres, count);
    // the very scene the context already holds (a host that sets it before every frame): nothing to do
    if (c->have_scene && count == c->scene.n_spheres && c->h_spheres.size() == count &&
        (count == 0 || memcmp(c->h_spheres.data(), spheres, (size_t)count * sizeof(rt_sphere)) == 0))
        return RT_OK;
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = ensure_scene_capacity(c, count);
    if (rc != RT_OK) return rc;
    c->is_light.assign(c->scene_cap, 0);
    c->h_spheres.assign(spheres, spheres + count);
    c->cost_valid = c->order_valid = false;
    c->pixel_cost_valid = c->deal_valid = false;
    rearm_probe(c);                     // a new scene: hierarchy or plain sweep is measured again
    rc = upload_spheres(c, 0, count, spheres, count, c->stream, true);
    if (rc != RT_OK) {
        c->have_scene = false;          // the tables are in an unknown state
        return rc;
    }
    c->have_scene = true;
    return RT_OK;
}

RT_API int rt_update_spheres_async(rt_ctx *c, uint32_t first, uint32_t count, const rt_sphere *spheres, void *hip_stream) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    if (!c->have_scene && !c->multi) return fail(RT_ERR_STATE, "rt_set_scene must precede rt_update_spheres_async");
    if (count > 0 && !spheres) return fail(RT_ERR_ARG, "spheres is null");
    if (c->multi) return rt::multi_update_spheres(c, first, count, spheres);
    const uint32_t n = c->scene.n_spheres;
    if (first > n || count > n - first) return fail(RT_ERR_ARG, "spheres [%u, %u) of a scene of %u", first, first + count, n);
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    if (count) memcpy(c->h_spheres.data() + first, spheres, (size_t)count * sizeof(rt_sphere));
    // the last frames' costs still predict this one (moving spheres): the order stays, and is sorted again from
    // fresh costs after a few changes
    if (++c->order_age >= 8) c->order_valid = c->deal_valid = false;
    rc = upload_spheres(c, first, count, spheres, n, (hipStream_t)hip_stream, false);
    if (rc == RT_OK) rearm_probe_if_changed(c);
    return rc;
}

RT_API int rt_set_camera(rt_ctx *c, const rt_camera *cam) {
    if (!c || !cam) return fail(RT_ERR_ARG, "null argument");
    if (c->multi) return rt::multi_set_camera(c, cam);
    if (!c->have_cam || memcmp(&c->cam, cam, sizeof *cam) != 0) {
        if (++c->order_age >= 8) c->order_valid = c->deal_valid = false;        // a moved camera: order and deal stay for a few frames, then are sorted again
    }
    c->cam = *cam;                      // a kernel argument: nothing to upload
    c->have_cam = true;
    return RT_OK;
}

RT_API int rt_set_mode(rt_ctx *c, int mode) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    bool ok = mode == RT_MODE_PARITY || mode == RT_MODE_FAST;
#if RT_DIAGNOSTICS
    // 100+k / 200+k: A/B instances of the parity / fast arithmetic (rt_debug.h; not part of the contract)
    int n_par = 0, n_fast = 0;
    (void)rt::parity_instances(&n_par);
    (void)rt::fast_instances(&n_fast);
    ok = ok || (mode >= 100 && mode < 100 + n_par) || (mode >= 200 && mode < 200 + n_fast);
#endif
    if (!ok) return fail(RT_ERR_ARG, "mode %d", mode);
    if (c->multi) return rt::multi_set_mode(c, mode);
    c->mode = mode;
    return RT_OK;
}

RT_API int rt_reset(rt_ctx *c) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    if (c->multi) return rt::multi_reset(c, false);
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    c->current_sample = 0;
    c->launches = 0;
    c->last_ms = 0.0;
    c->seeds_default = false;
    c->pixels_current = true;
    return restore_state(c);
}

RT_API int rt_reset_async(rt_ctx *c, void *hip_stream) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    if (c->multi) return rt::multi_reset(c, true);
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = chain(c, (hipStream_t)hip_stream);
    if (rc != RT_OK) return rc;
    hipLaunchKernelGGL(rt_zero_counters_kernel, dim3(1), dim3(256), 0, (hipStream_t)hip_stream, c->d_counters, c->d_stats);
    HIP_TRY(hipGetLastError());
    c->seeds_default = true;            // the next launch reads d_seeds0
    c->current_sample = 0;
    c->launches = 0;
    c->last_ms = 0.0;
    return RT_OK;
}

RT_API int rt_render_async(rt_ctx *c, int n_samples, void *hip_stream) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    if (c->multi) return rt::multi_render(c, nullptr, n_samples, false);
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    if (!c->throttle_on || n_samples <= 0) return launch(c, n_samples, (hipStream_t)hip_stream);
    // rt_throttle's bookkeeping: this launch between two events of the ring (the oldest entry is waited for first)
    hipStream_t stream = (hipStream_t)hip_stream;
    rt_ctx::Flight &f = c->flight[c->flight_next];
    if (f.pending) {
        HIP_TRY(hipEventSynchronize(f.stop));
        f.pending = false;
    }
    rc = chain(c, stream);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipEventRecord(f.start, stream));
    rc = launch(c, n_samples, stream);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipEventRecord(f.stop, stream));
    f.n_samples = n_samples;
    f.pending = true;
    c->flight_next = (c->flight_next + 1) % rt_ctx::kFlights;
    return RT_OK;
}

RT_API int rt_throttle(rt_ctx *c, int max_in_flight, double *ms_per_pass) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    if (max_in_flight < 0) return fail(RT_ERR_ARG, "max_in_flight %d", max_in_flight);
    if (ms_per_pass) *ms_per_pass = 0.0;
    if (c->multi) {
        if (max_in_flight > 0) return RT_OK;
        const int n = rt::multi_shards(c);
        for (int r = 0; r < n; ++r) {
            rt_ctx *s = rt::multi_shard(c, r);
            HIP_TRY(hipSetDevice(s->device));
            HIP_TRY(hipStreamSynchronize(s->last_stream));
        }
        return rt::multi_wait_frame(c);
    }
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    if (!c->throttle_on) {
        for (auto &f : c->flight) {
            HIP_TRY(hipEventCreate(&f.start));
            HIP_TRY(hipEventCreate(&f.stop));
        }
        c->throttle_on = true;
        if (max_in_flight == 0) return wait_all(c);
        return RT_OK;
    }
    // oldest first: retire what has finished, wait for the oldest while too many are left
    int pending = 0;
    for (const auto &f : c->flight) pending += f.pending ? 1 : 0;
    for (int k = 0; k < rt_ctx::kFlights && pending > 0; ++k) {
        rt_ctx::Flight &f = c->flight[(c->flight_next + k) % rt_ctx::kFlights];
        if (!f.pending) continue;
        if (pending > max_in_flight) {
            HIP_TRY(hipEventSynchronize(f.stop));
        } else if (hipEventQuery(f.stop) != hipSuccess) {
            (void)hipGetLastError();
            break;                          // (launches of one context finish in issue order)
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, f.start, f.stop) == hipSuccess && f.n_samples > 0) c->flight_ms_per_pass = (double)ms / f.n_samples;
        else (void)hipGetLastError();
        f.pending = false;
        pending -= 1;
    }
    if (max_in_flight == 0) {
        rc = wait_all(c);                   // (launches from before the first rt_throttle call, copies)
        if (rc != RT_OK) return rc;
    }
    if (ms_per_pass) *ms_per_pass = c->flight_ms_per_pass;
    return RT_OK;
}

RT_API int rt_render_pass(rt_ctx *c, uint32_t *out_host, int n_samples) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    if (c->multi) return rt::multi_render(c, out_host, n_samples, true);
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = chain(c, c->stream);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    rc = launch(c, n_samples, c->stream, true);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    if (out_host && c->local_rows > 0)
        HIP_TRY(hipMemcpyAsync(out_host, c->d_pixels_ext ? c->d_pixels_ext : c->d_pixels,
                               (size_t)c->local_rows * c->w * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_ms = ms;
    return RT_OK;
}

RT_API int rt_set_pixel_write(rt_ctx *c, int enable) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    if (c->multi) return rt::multi_set_pixel_write(c, enable);
    c->pixel_write = enable ? 1 : 0;
    return RT_OK;
}

RT_API int rt_read_pixels(rt_ctx *c, uint32_t *out_host) {
    if (!c || !out_host) return fail(RT_ERR_ARG, "null argument");
    if (c->multi) return rt::multi_read_pixels(c, out_host);
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    if (c->local_rows == 0) return RT_OK;
    rc = chain(c, c->stream);
    if (rc != RT_OK) return rc;
    if (!c->pixels_current && c->current_sample > 0) {
        rt::LaunchParams p = make_params(c, 0);
        hipError_t e = (c->mode == RT_MODE_FAST || c->mode >= 200) ? rt::launch_pack_fast(p, c->stream) : rt::launch_pack_parity(p, c->stream);
        if (e != hipSuccess) return fail(RT_ERR_HIP, "pack kernel launch failed: %s", hipGetErrorString(e));
        c->pixels_current = true;
    }
    HIP_TRY(hipMemcpyAsync(out_host, c->d_pixels_ext ? c->d_pixels_ext : c->d_pixels,
                           (size_t)c->local_rows * c->w * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

RT_API int rt_read_pixels_async(rt_ctx *c, uint32_t *out_host, void *hip_stream) {
    if (!c || !out_host) return fail(RT_ERR_ARG, "null argument");
    if (c->multi) return rt::multi_read_pixels(c, out_host);
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    if (c->local_rows == 0) return RT_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    rc = chain(c, stream);
    if (rc != RT_OK) return rc;
    if (!c->pixels_current && c->current_sample > 0) {
        rt::LaunchParams p = make_params(c, 0);
        hipError_t e = (c->mode == RT_MODE_FAST || c->mode >= 200) ? rt::launch_pack_fast(p, stream) : rt::launch_pack_parity(p, stream);
        if (e != hipSuccess) return fail(RT_ERR_HIP, "pack kernel launch failed: %s", hipGetErrorString(e));
        c->pixels_current = true;
    }
    HIP_TRY(hipMemcpyAsync(out_host, c->d_pixels_ext ? c->d_pixels_ext : c->d_pixels, (size_t)c->local_rows * c->w * sizeof(uint32_t),
                           hipMemcpyDeviceToHost, stream));
    return RT_OK;
}

RT_API int rt_pin_output(rt_ctx *c, uint32_t *out_host, size_t count) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    if (c->multi) return rt::multi_pin_output(c, out_host, count);
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    if (c->pinned_out) {
        rc = wait_all(c);
        if (rc != RT_OK) return rc;
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

RT_API int rt_set_pixel_buffer(rt_ctx *c, void *dptr, size_t count) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    if (c->multi) return fail(RT_ERR_ARG, "rt_set_pixel_buffer: a multi-device context assembles its frame in its own buffer");
    if (dptr && count < (size_t)c->local_rows * (size_t)c->w)
        return fail(RT_ERR_ARG, "pixel buffer of %zu < %zu elements", count, (size_t)c->local_rows * (size_t)c->w);
    c->d_pixels_ext = static_cast<uint32_t *>(dptr);
    return RT_OK;
}

// The context's own stream (hipStream_t, non-blocking): what rt_render_pass launches on.  Callers
// that keep several contexts in flight can launch each on its own stream through
// rt_render_async(ctx, n, rt_stream(ctx)).
RT_API void *rt_stream(rt_ctx *c) { return !c ? nullptr : (c->multi ? rt::multi_stream(c) : (void *)c->stream); }

RT_API int rt_device_pixels(rt_ctx *c, void **dptr, size_t *count) {
    if (!c || !dptr || !count) return fail(RT_ERR_ARG, "null argument");
    if (c->multi) return rt::multi_device_pixels(c, dptr, count);
    *dptr = c->d_pixels_ext ? c->d_pixels_ext : c->d_pixels;
    *count = (size_t)c->local_rows * (size_t)c->w;
    return RT_OK;
}

RT_API int rt_local_rows(const rt_ctx *c) { return c ? c->local_rows : RT_ERR_ARG; }
RT_API int rt_current_sample(const rt_ctx *c) { return c ? c->current_sample : RT_ERR_ARG; }

RT_API int rt_read_colors(rt_ctx *c, float *out_host) {
    if (!c || !out_host) return fail(RT_ERR_ARG, "null argument");
    if (c->multi) return rt::multi_read_colors(c, out_host);
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = chain(c, c->stream);           // behind everything the context has queued, on whatever stream
    if (rc != RT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out_host, c->d_colors, 3 * (size_t)c->w * c->h * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

RT_API int rt_read_seeds(rt_ctx *c, uint32_t *out_host) {
    if (!c || !out_host) return fail(RT_ERR_ARG, "null argument");
    if (c->multi) return rt::multi_read_seeds(c, out_host);
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = chain(c, c->stream);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out_host, c->seeds_default ? c->d_seeds0 : c->d_seeds, 2 * (size_t)c->w * c->h * sizeof(uint32_t),
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

RT_API int rt_get_stats(rt_ctx *c, rt_stats *out) {
    if (!c || !out) return fail(RT_ERR_ARG, "null argument");
    if (c->multi) return rt::multi_get_stats(c, out);
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = chain(c, c->stream);
    if (rc != RT_OK) return rc;
    unsigned long long v[32];
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

RT_API int rt_deinterleave_rows(uint32_t *full, const uint32_t *gathered, int w, int h, int nranks, int tile_rows, int pad_rows,
                                int device, void *hip_stream) {
    if (!full || !gathered) return fail(RT_ERR_ARG, "null argument");
    if (w <= 0 || h <= 0 || nranks < 1 || tile_rows <= 0) return fail(RT_ERR_ARG, "image %dx%d, %d ranks, tiles of %d rows", w, h, nranks, tile_rows);
    const int n_tiles = (h + tile_rows - 1) / tile_rows, need = ((n_tiles + nranks - 1) / nranks) * tile_rows;
    if (pad_rows < need && pad_rows < ((n_tiles - 1) / nranks) * tile_rows + (h - (n_tiles - 1) * tile_rows))
        return fail(RT_ERR_ARG, "pad_rows %d is less than a rank's row count", pad_rows);
    HIP_TRY(hipSetDevice(device));
    const int vec = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(full) | reinterpret_cast<uintptr_t>(gathered)) % 16 == 0);
    const size_t total = (size_t)(vec ? w / 4 : w) * (size_t)h;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(rt_deinterleave_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, full, gathered, w, h, nranks,
                       tile_rows, pad_rows, vec);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// ---- rt_render: the one-shot call, with the device state of recent sizes kept ----------------
namespace {
struct CacheEntry {
    int w = 0, h = 0;
    rt_ctx *ctx = nullptr;
    uint32_t *h_pix = nullptr;          // page-locked, w*h
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
    unsigned long long stamp = 0;
};
constexpr int kCacheSlots = 4;
CacheEntry g_cache[kCacheSlots];
unsigned long long g_cache_clock = 0;
std::mutex g_cache_mu;

void cache_drop(CacheEntry &e) {
    if (e.ctx) {
        (void)hipSetDevice(e.ctx->device);
        rt_destroy(e.ctx);
    }
    if (e.h_pix) (void)hipHostFree(e.h_pix);
    for (auto &v : e.ev)
        if (v) (void)hipEventDestroy(v);
    e = CacheEntry{};
}

int cache_get(int w, int h, CacheEntry **out) {
    int victim = 0;
    for (int k = 0; k < kCacheSlots; ++k) {
        if (g_cache[k].ctx && g_cache[k].w == w && g_cache[k].h == h) {
            *out = &g_cache[k];
            return RT_OK;
        }
        if (g_cache[k].stamp < g_cache[victim].stamp) victim = k;
    }
    CacheEntry &e = g_cache[victim];
    cache_drop(e);
    int rc = rt_create(&e.ctx, w, h);
    if (rc != RT_OK) return rc;
    e.w = w;
    e.h = h;
    hipError_t he = hipHostMalloc(reinterpret_cast<void **>(&e.h_pix), (size_t)w * h * sizeof(uint32_t), hipHostMallocDefault);
    for (int k = 0; k < 4 && he == hipSuccess; ++k) he = hipEventCreateWithFlags(&e.ev[k], hipEventDisableTiming);
    if (he != hipSuccess) {
        cache_drop(e);
        return fail(RT_ERR_ALLOC, "rt_render staging: %s", hipGetErrorString(he));
    }
    *out = &e;
    return RT_OK;
}
}  // namespace

RT_API void rt_release_cache(void) {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    for (auto &e : g_cache) cache_drop(e);
}

RT_API int rt_render(const rt_scene *scene, const rt_camera *cam, uint32_t *out, int w, int h, int spp) {
    if (!scene || !cam || !out) return fail(RT_ERR_ARG, "null argument");
    if (spp < 0) return fail(RT_ERR_ARG, "spp < 0");
    if (w <= 0 || h <= 0) return fail(RT_ERR_ARG, "image size %dx%d", w, h);
    std::lock_guard<std::mutex> lock(g_cache_mu);
    CacheEntry *e = nullptr;
    int rc = cache_get(w, h, &e);
    if (rc != RT_OK) return rc;
    e->stamp = ++g_cache_clock;
    rt_ctx *c = e->ctx;
    auto run = [&]() -> int {
        // a fresh OpenCLConfigBuffer: pass 0, default seed stream (read in place), parity mode, own pixel buffer
        int r = rt_reset_async(c, c->stream);
        if (r == RT_OK) r = rt_set_mode(c, RT_MODE_PARITY);
        if (r == RT_OK) r = rt_set_pixel_write(c, 1);
        if (r == RT_OK) r = rt_set_pixel_buffer(c, nullptr, 0);
        if (r == RT_OK) r = rt_set_scene(c, scene->spheres, scene->count);
        if (r == RT_OK) r = rt_set_camera(c, cam);
        if (r == RT_OK) r = launch(c, spp, c->stream, true);        // (blocking call: a new large scene may be probed first)
        if (r != RT_OK) return r;
        if (spp == 0) {                 // no pass ran: getPixels() of a fresh backend is the zero-filled buffer
            HIP_TRY(hipStreamSynchronize(c->stream));
            memset(out, 0, (size_t)w * h * sizeof(uint32_t));
            return RT_OK;
        }
        // read-back in four row blocks through page-locked staging: block k+1 crosses the bus while block k is
        // copied into the caller's (pageable) buffer
        const size_t total = (size_t)w * h;
        size_t off[5];
        for (int k = 0; k <= 4; ++k) off[k] = ((size_t)h * k / 4) * (size_t)w;
        for (int k = 0; k < 4; ++k) {
            if (off[k + 1] > off[k])
                HIP_TRY(hipMemcpyAsync(e->h_pix + off[k], c->d_pixels + off[k], (off[k + 1] - off[k]) * sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipEventRecord(e->ev[k], c->stream));
        }
        for (int k = 0; k < 4; ++k) {
            HIP_TRY(hipEventSynchronize(e->ev[k]));
            if (off[k + 1] > off[k]) memcpy(out + off[k], e->h_pix + off[k], (off[k + 1] - off[k]) * sizeof(uint32_t));
        }
        (void)total;
        return RT_OK;
    };
    rc = run();
    if (rc != RT_OK) {
        char keep[sizeof g_err];
        memcpy(keep, g_err, sizeof keep);
        cache_drop(*e);                 // never reuse a context that failed half-way
        memcpy(g_err, keep, sizeof keep);
    }
    return rc;
}

#if RT_DIAGNOSTICS
// =============================== diagnostics build only (rt_debug.h) ===========================

// host milliseconds of the last rt_create / rt_create_sharded of this process, by phase (g_create_ms above)
RT_API int rt_debug_create_breakdown(double *out8) {
    if (!out8) return fail(RT_ERR_ARG, "null argument");
    memcpy(out8, g_create_ms, sizeof g_create_ms);
    return RT_OK;
}
RT_API int rt_debug_variant_count(int fast) {
    int n = 0;
    (void)instances(fast != 0, &n);
    return n;
}
// the rt_set_mode value that selects the instance with this kernel symbol (100 + row / 200 + row), or RT_ERR_ARG
RT_API int rt_debug_instance(const char *name) {
    if (!name) return fail(RT_ERR_ARG, "name is null");
    for (int fast = 0; fast < 2; ++fast) {
        int n = 0;
        const rt::Instance *t = instances(fast != 0, &n);
        for (int k = 0; k < n; ++k)
            if (strcmp(t[k].name, name) == 0) return (fast ? 200 : 100) + k;
    }
    return fail(RT_ERR_ARG, "no instance named %s in this library", name);
}
// the kernel instance the last launch of shard `shard` of a multi-device context used ("" for a plain context or beyond the shards)
RT_API const char *rt_debug_shard_kernel(rt_ctx *c, int shard) {
    if (!c || !c->multi || shard < 0 || shard >= rt::multi_shards(c)) return "";
    return rt::multi_shard(c, shard)->last_kernel;
}
// The render kernels' table staging alone (rt_stage_probe_kernel), `repeats` launches of the grid and workgroup shape the library
// would use for `n_samples` passes of the current scene: for a profiler run that isolates the L2 behaviour of those reads.
RT_API int rt_debug_stage_tables(rt_ctx *c, int n_samples, int repeats) {
    if (!c || c->multi || !c->have_scene) return fail(RT_ERR_ARG, "null / multi-device context, or no scene");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = chain(c, c->stream);
    if (rc != RT_OK) return rc;
    const size_t lds_all = rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, true, n_samples);
    const int mat = lds_all <= (size_t)c->mat_lds_limit ? 1 : 0;
    const size_t lds = rt::lds_bytes(c->scene.n_spheres, c->scene.n_lights, mat != 0, n_samples);
    if (lds > kLdsMax) return fail(RT_ERR_ARG, "tables of %zu B do not fit LDS", lds);
    const bool coop = c->coop_min > 0 && c->scene.n_spheres >= (uint32_t)c->coop_min;
    const bool w1 = lds + (coop ? 1536u : 256u) <= 6 * 1024;
    const int tile_w = w1 ? 8 : 32;
    const dim3 grid((unsigned)((c->w + tile_w - 1) / tile_w), (unsigned)((c->local_rows + 7) / 8));
    uint32_t *sink = reinterpret_cast<uint32_t *>(c->d_tile_cost);          // (scratch: n_tiles >= 1024 words are not needed -- index & 1023 of a buffer that large)
    if (!sink || c->n_tiles < 1024) return fail(RT_ERR_ARG, "image too small for the probe");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rt_stage_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_debug_stage_tables: %s", hipGetErrorString(e));
    for (int k = 0; k < repeats; ++k) {
        hipLaunchKernelGGL(rt_stage_probe_kernel, grid, dim3(w1 ? 64 : 256), lds, c->stream, c->scene, mat, sink);
        HIP_TRY(hipGetLastError());
    }
    c->cost_valid = c->order_valid = false;                                 // (the probe scribbled over the tile costs)
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}
// failure injection: the state a failed gather (ncclGroupEnd) leaves a multi-device context in -- every later call is refused
RT_API int rt_debug_break_gather(rt_ctx *c) {
    if (!c || !c->multi) return fail(RT_ERR_ARG, "not a multi-device context");
    return rt::multi_debug_break(c);
}
// the kernel symbol of row `row` of the parity (fast = 0) or fast table, or "" beyond it
RT_API const char *rt_debug_instance_name(int fast, int row) {
    int n = 0;
    const rt::Instance *t = instances(fast != 0, &n);
    return (row >= 0 && row < n) ? t[row].name : "";
}

static int dbg_set_gate(rt_ctx *c, int v) { c->regen_gate = v; return RT_OK; }
static int dbg_set_matlds(rt_ctx *c, int v) { c->mat_lds_limit = v; return RT_OK; }
static int dbg_set_persist(rt_ctx *c, int v) { c->persist = v ? 1 : 0; return RT_OK; }
static int dbg_set_ncus(rt_ctx *c, int v) { c->n_cus = v; return RT_OK; }
static int dbg_set_coop(rt_ctx *c, int v) { c->coop_min = v & 0xffffff; c->coop_kmax = v >> 24; return RT_OK; }
static int dbg_set_wg(rt_ctx *c, int v) { c->wg_waves = v; return RT_OK; }
static int dbg_set_order(rt_ctx *c, int v) { c->use_order = v ? 1 : 0; if (v >> 8) c->order_homes = v >> 8; c->order_valid = false; return RT_OK; }
static int dbg_set_deal(rt_ctx *c, int v) {       // 0 = off; rows of a region | pixels of a run << 8
    c->use_deal = v ? 1 : 0;
    if (v & 255) c->deal_rows = v & 255;
    if (v >> 8) c->deal_group = v >> 8;
    c->deal_valid = false;
    c->cost_valid = c->order_valid = false;
    return RT_OK;
}
static int dbg_apply(rt_ctx *c, int (*fn)(rt_ctx *, int), int v) { return c->multi ? rt::multi_debug_each(c, fn, v) : fn(c, v); }

// tuning knob (not part of the contract): 0 = automatic, 1 = free-running, n = gate of n lanes
RT_API int rt_debug_set_regen_gate(rt_ctx *c, int gate) {
    if (!c || gate < 0 || gate > 64) return fail(RT_ERR_ARG, "gate %d", gate);
    return dbg_apply(c, dbg_set_gate, gate);
}
RT_API int rt_debug_set_mat_lds_limit(rt_ctx *c, int bytes) {
    if (!c || bytes < 0) return fail(RT_ERR_ARG, "bytes %d", bytes);
    return dbg_apply(c, dbg_set_matlds, bytes);
}
RT_API int rt_debug_set_persist(rt_ctx *c, int on) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    return dbg_apply(c, dbg_set_persist, on);
}
RT_API int rt_debug_set_ncus(rt_ctx *c, int n) {      // shrink the persistent grid (tests of the tile queue)
    if (!c || n < 1) return fail(RT_ERR_ARG, "n %d", n);
    return dbg_apply(c, dbg_set_ncus, n);
}
RT_API int rt_debug_set_wg_waves(rt_ctx *c, int waves) {    // 0 = automatic, 1 = single-wavefront workgroups, 4 = four wavefronts
    if (!c || (waves != 0 && waves != 1 && waves != 4)) return fail(RT_ERR_ARG, "waves %d", waves);
    return dbg_apply(c, dbg_set_wg, waves);
}
RT_API int rt_debug_set_tile_order(rt_ctx *c, int on) {      // 0: tiles in their natural order (the round-1 behaviour); 1: heavy first; | homes << 8 (1 .. 8; 1 = regions not kept on one XCD)
    if (!c || on < 0 || (on >> 8) > 8) return fail(RT_ERR_ARG, "ctx is null / order %d", on);
    return dbg_apply(c, dbg_set_order, on);
}
RT_API int rt_debug_set_pixel_deal(rt_ctx *c, int rows) {     // 0: every wavefront renders its 8x8 square (the round-2 behaviour); else rows of a region (8 .. 128) | pixels of a run (1, 2, 4, 8; 0 = keep) << 8
    const int r = rows & 255, g = rows >> 8;
    if (!c || rows < 0 || (rows != 0 && r != 8 && r != 16 && r != 32 && r != 64 && r != 128) || (g != 0 && g != 1 && g != 2 && g != 4 && g != 8))
        return fail(RT_ERR_ARG, "rows %d, run %d", r, g);
    return dbg_apply(c, dbg_set_deal, rows);
}
// the deal in use (valid = 0: none) -- per region 256 positions dy * 32 + dx in rank order -- and the per-pixel costs of the last launch
RT_API int rt_debug_read_pixel_deal(rt_ctx *c, uint16_t *deal_out, size_t deal_cap, uint16_t *cost_out, size_t cost_cap, int *valid) {
    if (!c || c->multi) return fail(RT_ERR_ARG, "null / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    const size_t regions = (size_t)((c->w + rt::kRegionW - 1) / rt::kRegionW) * (size_t)((c->local_rows + c->deal_rows - 1) / c->deal_rows);
    const size_t per_region = (size_t)rt::kRegionW * c->deal_rows;
    const size_t n_deal = regions * per_region < deal_cap ? regions * per_region : deal_cap, n_cost = (size_t)c->local_rows * c->w < cost_cap ? (size_t)c->local_rows * c->w : cost_cap;
    if (deal_out && n_deal && c->d_deal) HIP_TRY(hipMemcpy(deal_out, c->d_deal, n_deal * sizeof(uint16_t), hipMemcpyDeviceToHost));
    if (cost_out && n_cost && c->d_pixel_cost) HIP_TRY(hipMemcpy(cost_out, c->d_pixel_cost, n_cost * sizeof(uint16_t), hipMemcpyDeviceToHost));
    if (valid) *valid = c->deal_valid ? c->deal_rows : 0;
    return RT_OK;
}
// the tile order in use (valid = 0: none, tiles run in their natural order) and the per-tile costs of the last launch
RT_API int rt_debug_read_tile_order(rt_ctx *c, uint32_t *order_out, uint32_t *cost_out, uint32_t cap, uint32_t *n_tiles, int *valid) {
    if (!c || c->multi) return fail(RT_ERR_ARG, "null / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    const uint32_t in_use = c->cost_valid && c->cost_tiles ? c->cost_tiles : c->n_tiles;      // (the tiles of the last launch's workgroup shape)
    const uint32_t n = cap < in_use ? cap : in_use;
    if (order_out && n) HIP_TRY(hipMemcpy(order_out, c->d_order, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (cost_out && n) HIP_TRY(hipMemcpy(cost_out, c->d_tile_cost, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (n_tiles) *n_tiles = in_use;
    if (valid) *valid = c->order_valid ? 1 : 0;
    return RT_OK;
}
// the hierarchy of large scenes: min_spheres = smallest tree that is built and used (0 = never), lds_limit = largest
// LDS footprint it is used at (0 = keep).  Takes effect at once: the current scene's tables are rebuilt.
static int dbg_set_bvh_lds(rt_ctx *c, int v) { if (v > 0) c->bvh_lds_limit = v; return RT_OK; }
static int dbg_set_bvh_min(rt_ctx *c, int v) {
    c->bvh_min = v;
    rearm_probe(c);
    if (!c->have_scene) return RT_OK;
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = chain(c, c->stream);
    return rc != RT_OK ? rc : rt::build_bvh(c, c->scene.n_spheres, c->stream, true);
}
static int dbg_set_tree_shape(rt_ctx *c, int v) {
    c->bvh_sah = v ? 1 : 0;
    return dbg_set_bvh_min(c, c->bvh_min);          // (rebuilds the current scene's tables, re-arms the probe)
}
// 1: full scene uploads build the hierarchy on the host with its shape chosen by surface area (the default); 0: the device build
// and its fixed shape for them too (what device-resident updates always use).  Takes effect at once.
RT_API int rt_debug_set_tree_shape(rt_ctx *c, int by_area) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    return dbg_apply(c, dbg_set_tree_shape, by_area);
}
static int dbg_set_walk_gate(rt_ctx *c, int v) { if (v > 0) c->walk_gate = v; return RT_OK; }
static int dbg_set_walk_round(rt_ctx *c, int v) { c->walk_round = v; return RT_OK; }

RT_API int rt_debug_set_walk_round(rt_ctx *c, int steps) {
    if (!c || steps < 1) return fail(RT_ERR_ARG, "steps %d", steps);
    return dbg_apply(c, dbg_set_walk_round, steps);
}

static int dbg_set_walk_forced(rt_ctx *c, int v) { c->walk_forced = v ? 1 : 0; rearm_probe(c); return RT_OK; }
// rt_walk.inc.h: pair steps per lane per loop trip, ready lanes that make a wavefront shade (0 = keep either), and
// forced: 0 = hierarchy or plain sweep by measurement (the library's behaviour), 1 = the hierarchy whenever the scene has one
RT_API int rt_debug_set_walk(rt_ctx *c, int steps, int gate, int forced) {
    if (!c || steps < 0 || gate < 0 || gate > 64 || forced < 0 || forced > 1) return fail(RT_ERR_ARG, "steps %d, gate %d, forced %d", steps, gate, forced);
    int rc = dbg_apply(c, dbg_set_walk_gate, gate);         // (`steps`: the per-trip step budget of rounds 2-3; a walk now runs to its end within the trip)
    return rc != RT_OK ? rc : dbg_apply(c, dbg_set_walk_forced, forced);
}
// rays8[i] = { o.xyz, t_max, d.xyz, shadow != 0 } through the hierarchy walk and through the plain sweep (csrc/rt_walk.inc.h
// rt_walk_rays kernel); out4[i] = the walk's answer, then the sweep's (closest: distance bits, scene index; shadow: first
// blocking index, 0)
RT_API int rt_debug_walk_rays(rt_ctx *c, const float *rays8, uint32_t n_rays, uint32_t *out4) {
    if (!c || c->multi || !rays8 || !out4) return fail(RT_ERR_ARG, "null / multi-device context");
    if (!c->bvh_ok) return fail(RT_ERR_STATE, "the scene has no hierarchy (rt_debug_set_bvh)");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    if (!c->have_cam) c->cam = rt_camera{};
    rt::LaunchParams p = make_params(c, 1);
    p.bvh = c->bvh;
    const size_t lds = rt::lds_bytes_pairs(0, 0, false, 0, c->bvh.n_leaves, c->bvh.n_slots, c->bvh.stack_depth, 256);
    if (lds > 152 * 1024) return fail(RT_ERR_ARG, "tables need %zu B of LDS", lds);
    float4 *d_rays = nullptr;
    uint4 *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_rays, (size_t)n_rays * 32 + 32));
    hipError_t e = hipMalloc(&d_out, (size_t)n_rays * 16 + 16);
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays8, (size_t)n_rays * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rt::launch_walk_rays(p, d_rays, n_rays, d_out, lds, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out4, d_out, (size_t)n_rays * 16, hipMemcpyDeviceToHost);
    (void)hipFree(d_rays);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_debug_walk_rays: %s", hipGetErrorString(e));
    return RT_OK;
}
// the estimate that settles hierarchy against sweep without a launch: out4 = { expected pair steps, expected leaf visits,
// predicted walk / sweep time per ray, 1 if the verdict in force came from it (0: measured, or none yet) }; returns 1 when
// the context holds an estimate for its scene.  rt_debug_set_choice_estimate(ctx, 0) switches it off: every undecided
// scene is then measured (the calibration's way of getting both timings).
RT_API int rt_debug_tree_estimate(rt_ctx *c, double *out4) {
    if (!c || c->multi || !out4) return fail(RT_ERR_ARG, "null / multi-device context");
    out4[0] = c->bvh_est_pairs;
    out4[1] = c->bvh_est_leaves;
    out4[2] = c->bvh_est_valid && c->bvh_ok ? estimate_ratio(c) : 0.0;
    out4[3] = c->pick_estimated ? 1.0 : 0.0;
    return c->bvh_est_valid && c->bvh_ok ? 1 : 0;
}
static int dbg_set_estimate(rt_ctx *c, int v) { c->use_estimate = v ? 1 : 0; rearm_probe(c); return RT_OK; }
RT_API int rt_debug_set_choice_estimate(rt_ctx *c, int on) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    return dbg_apply(c, dbg_set_estimate, on);
}
RT_API int rt_debug_bvh_pick(rt_ctx *c) {
    if (!c || c->multi) return fail(RT_ERR_ARG, "null / multi-device context");
    if (select_device(c) == RT_OK) probe_poll(c, false);
    return c->bvh_pick;
}
RT_API int rt_debug_set_bvh(rt_ctx *c, int min_spheres, int lds_limit) {
    if (!c || min_spheres < 0 || lds_limit < 0 || lds_limit > 152 * 1024) return fail(RT_ERR_ARG, "min_spheres %d, lds_limit %d", min_spheres, lds_limit);
    int rc = dbg_apply(c, dbg_set_bvh_lds, lds_limit);
    return rc != RT_OK ? rc : dbg_apply(c, dbg_set_bvh_min, min_spheres);
}
// the blob of rt_device.h BvhTables as it lies in HBM (float4 units), and four numbers {always, leaves, stack depth, root pair}
// (slots = always + 8 * leaves);
// counts of 0 = the scene has no hierarchy
RT_API int rt_debug_read_bvh(rt_ctx *c, float *blob_out, uint32_t cap_float4, uint32_t *counts4) {
    if (!c || c->multi || !counts4) return fail(RT_ERR_ARG, "null / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    counts4[0] = counts4[1] = counts4[2] = counts4[3] = 0;
    if (!c->bvh_ok) return RT_OK;
    counts4[0] = c->bvh.n_always; counts4[1] = c->bvh.n_leaves; counts4[2] = c->bvh.stack_depth; counts4[3] = c->bvh.root;
    const size_t need = rt::bvh_blob_float4s(c->bvh.n_leaves, c->bvh.n_slots);
    if (blob_out) {
        if (cap_float4 < need) return fail(RT_ERR_ARG, "blob needs %zu float4", need);
        HIP_TRY(hipMemcpy(blob_out, c->d_bvh, need * sizeof(float4), hipMemcpyDeviceToHost));
    }
    return RT_OK;
}
// min_spheres | kmax << 24: scenes of at least min_spheres use the cooperative any-hit instance; kmax: it shares a sweep out only while
// no more than kmax shadow rays are pending in the wavefront (0 = no limit)
RT_API int rt_debug_set_coop_min(rt_ctx *c, int min_spheres) {
    if (!c || min_spheres < 0) return fail(RT_ERR_ARG, "min_spheres %d", min_spheres);
    return dbg_apply(c, dbg_set_coop, min_spheres);
}

// section cycle sums of a stamped instance (valid after rt_get_stats)
RT_API int rt_debug_counters(rt_ctx *c, unsigned long long *out24) {
    if (!c || !out24 || c->multi) return fail(RT_ERR_ARG, "null argument / multi-device context");
    memcpy(out24, c->debug_counters, sizeof c->debug_counters);
    return RT_OK;
}

RT_API int rt_debug_counters_raw(rt_ctx *c, unsigned long long *out32) {
    if (!c || !out32 || c->multi) return fail(RT_ERR_ARG, "null argument / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = chain(c, c->stream);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out32, c->d_counters, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

static unsigned long long *timelog_next(rt_ctx *c, uint32_t *seq_out) {
    if (!c->d_timelog || c->timelog_used >= c->timelog_cap) return nullptr;
    *seq_out = c->timelog_used;
    return c->d_timelog + 8 * (size_t)(c->timelog_used++);
}

// NOTE: deliberately NOT chained (no event dependency added by the library): this is the round-1 reset,
// kept to reproduce and study the ordering failure recorded in DESIGN.md section 3
RT_API int rt_debug_reset_by_copy(rt_ctx *c, void *hip_stream, int flags) {
    if (!c || c->multi) return fail(RT_ERR_ARG, "null / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    const size_t n = 2 * (size_t)c->w * (size_t)c->h;
    if (flags & 1) {
        HIP_TRY(hipMemcpyAsync(c->d_seeds, c->d_seeds0, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
    } else {
        uint32_t seq = 0;
        unsigned long long *tl = timelog_next(c, &seq);
        unsigned long long *bl = (tl && c->d_blocklog) ? c->d_blocklog + (size_t)seq * 1024 : nullptr;
        hipLaunchKernelGGL(rt_debug_copy_seeds_kernel, dim3(1024), dim3(256), 0, (hipStream_t)hip_stream, c->d_seeds, c->d_seeds0, n, tl,
                           c->timelog_tag, bl, flags);
        HIP_TRY(hipGetLastError());
    }
    c->seeds_default = false;
    c->current_sample = 0;
    return RT_OK;
}

// a kernel on `hip_stream` that counts the seed words differing from the default stream
// into counters[28] (and the number of probes into counters[29]); read them with rt_debug_counters_raw
RT_API int rt_debug_probe_seeds(rt_ctx *c, void *hip_stream, int flags) {
    if (!c || c->multi) return fail(RT_ERR_ARG, "null / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    uint32_t seq = 0;
    unsigned long long *tl = timelog_next(c, &seq);
    uint32_t *sl = (tl && c->d_stalelog) ? c->d_stalelog + (size_t)seq * 64 : nullptr;
    hipLaunchKernelGGL(rt_debug_probe_seeds_kernel, dim3(256), dim3(256), 0, (hipStream_t)hip_stream, c->d_seeds, c->d_seeds0,
                       2 * (size_t)c->w * (size_t)c->h, c->d_counters + 28, tl, c->timelog_tag, sl, flags);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

RT_API int rt_debug_timelog_enable(rt_ctx *c, uint32_t entries, uint32_t wave_entries) {
    if (!c || c->multi) return fail(RT_ERR_ARG, "null / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    (void)hipFree(c->d_timelog);
    (void)hipFree(c->d_wavelog);
    (void)hipFree(c->d_blocklog);
    (void)hipFree(c->d_stalelog);
    c->d_timelog = c->d_wavelog = c->d_blocklog = nullptr;
    c->d_stalelog = nullptr;
    c->timelog_cap = c->timelog_used = c->wavelog_cap = 0;
    if (entries) {
        HIP_TRY(hipMalloc(&c->d_blocklog, (size_t)entries * 1024 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(c->d_blocklog, 0, (size_t)entries * 1024 * sizeof(unsigned long long)));
        if (wave_entries == 0xC0FFEEu) {
            // provenance experiment (tools/gather_stress.py RT_LOG_PATTERN=1): the zeros above were written by the
            // runtime's fill KERNEL (shader stores through some XCD's L2); now the same bytes are overwritten with a
            // pattern by a host-to-device copy that does not go through any L2.  A workgroup log entry that is later
            // found lost then tells by its value what happened: the pattern = the write never arrived; zero = a stale
            // line from the fill kernel was written back over it afterwards.
            HIP_TRY(hipDeviceSynchronize());
            std::vector<unsigned long long> pat((size_t)entries * 1024, 0x5555555555555550ull);
            HIP_TRY(hipMemcpy(c->d_blocklog, pat.data(), pat.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
            HIP_TRY(hipDeviceSynchronize());
            wave_entries = 0;
        }
        HIP_TRY(hipMalloc(&c->d_stalelog, (size_t)entries * 64 * sizeof(uint32_t)));
        HIP_TRY(hipMemset(c->d_stalelog, 0, (size_t)entries * 64 * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->d_timelog, (size_t)entries * 8 * sizeof(unsigned long long)));
        std::vector<unsigned long long> init((size_t)entries * 8, 0ull);
        for (uint32_t i = 0; i < entries; ++i) init[8 * (size_t)i] = ~0ull;
        HIP_TRY(hipMemcpy(c->d_timelog, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        c->timelog_cap = entries;
    }
    if (wave_entries) {
        HIP_TRY(hipMalloc(&c->d_wavelog, (size_t)wave_entries * 3 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(c->d_wavelog, 0, (size_t)wave_entries * 3 * sizeof(unsigned long long)));
        c->wavelog_cap = wave_entries;
    }
    return RT_OK;
}

RT_API int rt_debug_timelog_tag(rt_ctx *c, unsigned long long tag) {
    if (!c) return fail(RT_ERR_ARG, "ctx is null");
    c->timelog_tag = tag;
    return RT_OK;
}

RT_API int rt_debug_timelog_read(rt_ctx *c, unsigned long long *out, uint32_t entries, uint32_t *used) {
    if (!c || !out || c->multi) return fail(RT_ERR_ARG, "null argument / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    const uint32_t n = entries < c->timelog_cap ? entries : c->timelog_cap;
    if (n) HIP_TRY(hipMemcpy(out, c->d_timelog, (size_t)n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (used) *used = c->timelog_used;
    return RT_OK;
}

// the per-workgroup log of copy record `seq` (1024 u64: start << 4 | xcc) and the stale-word log of probe
// record `seq` (64 u32: count, then index | reader xcc << 28)
RT_API int rt_debug_sidelog_read(rt_ctx *c, uint32_t seq, unsigned long long *blocklog1024, uint32_t *stalelog64) {
    if (!c || c->multi || seq >= c->timelog_cap) return fail(RT_ERR_ARG, "null / multi-device context / record %u", seq);
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    if (blocklog1024) HIP_TRY(hipMemcpy(blocklog1024, c->d_blocklog + (size_t)seq * 1024, 1024 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (stalelog64) HIP_TRY(hipMemcpy(stalelog64, c->d_stalelog + (size_t)seq * 64, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

RT_API int rt_debug_wavelog_read(rt_ctx *c, unsigned long long *out, uint32_t wave_entries) {
    if (!c || !out || c->multi) return fail(RT_ERR_ARG, "null argument / multi-device context");
    int rc = select_device(c);
    if (rc != RT_OK) return rc;
    rc = wait_all(c);
    if (rc != RT_OK) return rc;
    const uint32_t n = wave_entries < c->wavelog_cap ? wave_entries : c->wavelog_cap;
    if (n) HIP_TRY(hipMemcpy(out, c->d_wavelog, (size_t)n * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RT_OK;
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
RT_API long long rt_debug_sqrt_mismatches(void) { return sqrt_check(0); }
// sphere test with the unchecked square root against the one with sqrtf, tiny discriminants
RT_API long long rt_debug_hitpost_mismatches(void) { return sqrt_check(1); }

// mismatches of the candidate lean reciprocals per input exponent: out[4][256]
RT_API int rt_debug_rcp_probe(unsigned long long *out1024) {
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

RT_API int rt_debug_eval(int op, const float *in_host, float *out_host, size_t n) {
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
#endif   // RT_DIAGNOSTICS

}  // extern "C"
