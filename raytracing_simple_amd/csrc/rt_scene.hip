// rt_scene.hip -- the scene a context holds on its device: the raw 44-byte records as the host hands them over
// (OpenCLConfig.cpp:720-747), the tables the kernels read (built from them ON the device), the page-locked staging ring
// that uploads go through, and the hierarchy of large scenes behind the tables (rt_bvh.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "rt_internal.h"

#ifndef RT_DIAGNOSTICS
#define RT_DIAGNOSTICS 0
#endif

using rt::fail;

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

namespace rt {

// ---- scene storage -------------------------------------------------------------------------

static bool light_test(const rt_sphere &s) { return !((s.e.x == 0.f) && (s.e.z == 0.f)); }   // .cl:135-138,266

void free_scene(rt_ctx *c) {
    (void)hipFree(c->d_spheres);
    (void)hipFree(c->d_tables);
    (void)hipFree(c->d_bvh);
    (void)hipFree(c->d_dup);
    if (c->h_dup_stage) (void)hipHostFree(c->h_dup_stage);
    c->d_dup = nullptr;
    c->h_dup_stage = nullptr;
    c->dup_stage_used = false;
    c->have_dups = false;
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
    uint8_t *nd = nullptr;
    if (e == hipSuccess) e = hipMalloc(&nd, (size_t)cap);
    if (e != hipSuccess) {
        (void)hipFree(ns);
        (void)hipFree(nt);
        (void)hipFree(nb);
        return fail(RT_ERR_ALLOC, "scene tables for %u spheres: %s", cap, hipGetErrorString(e));
    }
    free_scene(c);
    c->d_spheres = ns;
    c->d_tables = nt;
    c->d_bvh = nb;
    c->d_dup = nd;
    c->scene_cap = cap;
    return RT_OK;
}

static int ensure_stage_capacity(rt_ctx *c, uint32_t count) {
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
static int build_tables(rt_ctx *c, uint32_t n_total, hipStream_t stream, bool full_upload);
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
    if (!full_upload && c->have_scene) {
        // a device-resident update: the records are on their way; the tables and the hierarchy are built from them when something reads them next
        // (refresh_tables: launch(), the diagnostics' readers) -- once for all the updates queued until then
        c->tables_stale = true;
        return RT_OK;
    }
    return build_tables(c, n_total, stream, full_upload);
}

// the scene tables (rt_build_tables_kernel) and the hierarchy from the records in c->d_spheres, on `stream`
static int build_tables(rt_ctx *c, uint32_t n_total, hipStream_t stream, bool full_upload) {
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
    const int rc = rt::build_bvh(c, n_total, stream, full_upload);
    if (rc == RT_OK) c->tables_stale = false;       // (a build that failed half-way is tried again by the next reader)
    return rc;
}

int refresh_tables(rt_ctx *c, hipStream_t stream) {
    if (!c->tables_stale || !c->have_scene) return RT_OK;
    int rc = chain(c, stream);          // (behind the copies of the records, whatever streams the updates were given)
    if (rc != RT_OK) return rc;
    rc = build_tables(c, c->scene.n_spheres, stream, false);
    if (rc == RT_OK) rearm_probe_if_changed(c);
    return rc;
}

}  // namespace rt
