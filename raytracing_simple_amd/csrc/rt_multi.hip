// rt_multi.hip -- one context on several GPUs of one process (rt_create(w,h,ngpus); 8e).
//
// The reference has no multi-device code; its host (SimpleRT/src/Main.cpp:29-102) drives ONE
// backend object through Config::updateRendering (Config.cpp:73-81) and reads ONE pixel buffer
// (OpenCLConfig.cpp:407-515 is what execute() must still look like to it).  A multi-device context
// keeps exactly that face: the image is sharded by interleaved row tiles (tile t -> device t % n),
// every device renders its rows with an ordinary sharded context on its own stream, and each frame
// ends with ONE gather into the first device over RCCL/xGMI --
//     ncclGroupStart;  root: ncclRecv x (n-1);  every other device: ncclSend;  ncclGroupEnd   (ncclUint32)
// -- seven point-to-point transfers over seven distinct links into the root, no ring; then the
// de-interleave kernel (rt_deinterleave_rows) and one copy to the host.  The root's own rows are
// rendered straight into its receive slot, so they are never copied.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the copy already in the process if there is one):
// hosts that never ask for a multi-device context do not load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include "rt_internal.h"

#ifndef RT_DIAGNOSTICS
#define RT_DIAGNOSTICS 0
#endif

using rt::fail;

namespace {

// ---- the slice of the RCCL API this file uses (rccl.h: ncclResult_t = int, ncclComm_t = opaque pointer) ----
typedef void *rcclComm;
struct Rccl {
    void *handle = nullptr;
    int (*CommInitAll)(rcclComm *, int, const int *) = nullptr;
    int (*CommDestroy)(rcclComm) = nullptr;
    int (*CommAbort)(rcclComm) = nullptr;         // optional: older libraries lack it
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, rcclComm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, rcclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
constexpr int kRcclUint32 = 3;          // ncclUint32 (rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3)

Rccl g_rccl;
std::mutex g_rccl_mu;
// Diagnostics build only (rt_debug_set_rccl_library): a library to bind INSTEAD of RCCL -- tests/rccl_double.cpp, which pairs
// sends with receives as device-to-device copies on the streams it is handed and can be told to fail a call -- and whether a
// device listed n times counts as n devices, so that the grouped send / receive branch and its failure handling run on one GPU.
std::string g_rccl_override;
bool g_repeated_counts_as_distinct = false;

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.handle) return RT_OK;
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void *h = nullptr;
    if (!g_rccl_override.empty()) {
        h = dlopen(g_rccl_override.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) return fail(RT_ERR_NO_DEVICE, "%s cannot be loaded (%s)", g_rccl_override.c_str(), dlerror());
    }
    for (const char *n : names)
        if (h || (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;             // a copy the process already has (torch's)
    for (const char *n : names) {
        if (h) break;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) return fail(RT_ERR_NO_DEVICE, "librccl.so.1 cannot be loaded (%s): a multi-device context needs RCCL", dlerror());
    Rccl r;
    r.handle = h;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(h, "ncclCommAbort"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(h, "ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(h, "ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv || !r.GetErrorString)
        return fail(RT_ERR_NO_DEVICE, "librccl.so.1 lacks one of ncclCommInitAll/ncclGroupStart/ncclSend/ncclRecv/...");
    g_rccl = r;
    return RT_OK;
}

#define RCCL_TRY(call)                                                                                  \
    do {                                                                                                \
        int r_ = (call);                                                                                \
        if (r_ != 0) return fail(RT_ERR_HIP, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), \
                                 __FILE__, __LINE__);                                                   \
    } while (0)

}  // namespace

struct rt_multi {
    int n = 0;
    int w = 0, h = 0, tile_rows = 8, pad_rows = 0;
    bool emulated = false;              // a device is listed twice: D2D copies stand in for ncclSend/ncclRecv
    std::vector<int> devices;
    std::vector<rt_ctx *> shard;        // shard[r] = rt_create_sharded(w, h, devices[r], r, n, tile_rows)
    std::vector<rcclComm> comm;         // one per device (ncclCommInitAll), empty when emulated
    // The gather of frame k runs on the root device's SECOND stream into receive slot k % 2, so that the root's render of
    // frame k + 1 (into the other slot) is not queued behind n - 1 receives and the de-interleave of frame k.
    uint32_t *d_gathered[2] = { nullptr, nullptr };   // root: [n][pad_rows][w] each, rank r's block at r * pad_rows * w
    uint32_t *d_full = nullptr;         // root: [h][w], the assembled frame (n = 1: the one shard renders straight into it)
    hipStream_t gather_stream = nullptr;
    hipEvent_t ev_rendered = nullptr;   // root's rows of the frame are in its slot (root's render stream)
    hipEvent_t ev_gathered[2] = { nullptr, nullptr };   // the slot's frame is assembled in d_full: the slot may be rendered into again
    bool slot_used[2] = { false, false };
    uint64_t frames = 0;                // frames gathered so far (the next one uses slot frames % 2)
    hipEvent_t ev_ready = nullptr;      // emulation: a shard's render done, on its device
    hipEvent_t ev_copied = nullptr;     // emulation: the root has copied the shards' rows out of their pixel buffers
    bool broken = false;                // a gather failed half-way: the communicators may hold an open or failed group
    char broken_why[256] = "";
    uint32_t *pinned_out = nullptr;
    double last_ms = 0.0;
    uint64_t launches = 0;
};

namespace {

int rows_of(int h, int rank, int n, int tile_rows) {
    const int n_tiles = (h + tile_rows - 1) / tile_rows;
    int rows = 0;
    for (int t = rank; t < n_tiles; t += n) rows += (t * tile_rows + tile_rows <= h) ? tile_rows : (h - t * tile_rows);
    return rows;
}

int refuse_if_broken(const rt_multi *m) {
    if (m->broken) return fail(RT_ERR_STATE, "this multi-device context is unusable since a gather failed (%s): destroy it", m->broken_why);
    return RT_OK;
}

// A gather that failed half-way may have posted a receive or a send whose partner never comes: whether ncclGroupEnd then
// returns the error or enqueues a transfer that never completes depends on the RCCL version.  So the communicators are
// ABORTED (ncclCommAbort tears down what is in flight on them), and the shards' streams -- which may hold such a transfer --
// are never synchronised again: rt_destroy skips the wait for a shard marked so.
int mark_broken(rt_multi *m, int code, const char *what, const char *detail) {
    m->broken = true;
    snprintf(m->broken_why, sizeof m->broken_why, "%s: %s", what, detail ? detail : "?");
    for (rcclComm &c : m->comm) {
        if (c && g_rccl.CommAbort) {
            (void)g_rccl.CommAbort(c);
            c = nullptr;
        }
    }
    for (rt_ctx *s : m->shard) s->abandon_streams = true;
    return fail(code, "%s failed: %s -- the multi-device context is now unusable", what, detail ? detail : "?");
}

// Before the shards render a frame that will be gathered: the root renders its own rows straight into the receive slot of
// that frame, which it may do once the slot's previous frame (two frames ago) has been assembled out of it.
int begin_frame(rt_multi *m) {
    if (m->n == 1) return RT_OK;
    const int slot = (int)(m->frames & 1u);
    rt_ctx *root = m->shard[0];
    HIP_TRY(hipSetDevice(m->devices[0]));
    if (m->slot_used[slot]) HIP_TRY(hipStreamWaitEvent(root->stream, m->ev_gathered[slot], 0));
    return rt_set_pixel_buffer(root, m->d_gathered[slot], (size_t)m->pad_rows * m->w);
}

// The frame-end gather + assembly, on the root device's gather stream: behind the root's own render (an event) and, through the
// pairing of ncclRecv with ncclSend, behind every other shard's.  The root's render stream is NOT touched: its next frame starts
// while this one is received and de-interleaved.
int gather_and_assemble(rt_multi *m) {
    int rc = refuse_if_broken(m);
    if (rc != RT_OK) return rc;
    if (m->n == 1) return RT_OK;                        // one device: its rows ARE the frame, rendered straight into d_full
    rt_ctx *root = m->shard[0];
    const int slot = (int)(m->frames & 1u);
    uint32_t *gathered = m->d_gathered[slot];
    const size_t block = (size_t)m->pad_rows * (size_t)m->w;
    // every transfer must fit the receive slot it lands in (slot r = pad_rows * w words at r * block)
    for (int r = 1; r < m->n; ++r) {
        const size_t count = (size_t)m->shard[r]->local_rows * (size_t)m->w;
        if (count > block || m->shard[r]->w != m->w)
            return fail(RT_ERR_STATE, "shard %d would send %zu words into a slot of %zu", r, count, block);
    }
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipEventRecord(m->ev_rendered, root->last_stream));
    HIP_TRY(hipStreamWaitEvent(m->gather_stream, m->ev_rendered, 0));
    if (!m->emulated) {
        // One group: root posts n-1 receives, every other device one send.  A call that fails inside the group must
        // not leave it open: the group is always ended, and any failure (there or in ncclGroupEnd) marks the context
        // unusable -- the communicators are aborted and later calls refused instead of queueing behind them.
        int r0 = g_rccl.GroupStart();
        if (r0 != 0) return mark_broken(m, RT_ERR_HIP, "ncclGroupStart", g_rccl.GetErrorString(r0));
        int bad = 0;
        const char *bad_what = "";
        hipError_t bad_hip = hipSuccess;
        for (int r = 1; r < m->n && !bad && bad_hip == hipSuccess; ++r) {
            const size_t count = (size_t)m->shard[r]->local_rows * (size_t)m->w;
            if (!count) continue;
            if ((bad_hip = hipSetDevice(m->devices[0])) != hipSuccess) break;
            if ((bad = g_rccl.Recv(gathered + (size_t)r * block, count, kRcclUint32, r, m->comm[0], m->gather_stream)) != 0) { bad_what = "ncclRecv"; break; }
            if ((bad_hip = hipSetDevice(m->devices[r])) != hipSuccess) break;
            if ((bad = g_rccl.Send(m->shard[r]->d_pixels, count, kRcclUint32, 0, m->comm[r], m->shard[r]->stream)) != 0) { bad_what = "ncclSend"; break; }
        }
        const int r1 = g_rccl.GroupEnd();
        if (bad) return mark_broken(m, RT_ERR_HIP, bad_what, g_rccl.GetErrorString(bad));
        if (bad_hip != hipSuccess) return mark_broken(m, RT_ERR_HIP, "hipSetDevice inside the gather", hipGetErrorString(bad_hip));
        if (r1 != 0) return mark_broken(m, RT_ERR_HIP, "ncclGroupEnd", g_rccl.GetErrorString(r1));
    } else {
        // one-GPU rehearsal: the receive slot is filled by a device-to-device copy on the gather stream, which first waits for
        // the sending shard's render (what ncclRecv's pairing with ncclSend does on real links)
        for (int r = 1; r < m->n; ++r) {
            const size_t count = (size_t)m->shard[r]->local_rows * (size_t)m->w;
            if (!count) continue;
            HIP_TRY(hipSetDevice(m->devices[r]));
            HIP_TRY(hipEventRecord(m->ev_ready, m->shard[r]->last_stream));
            HIP_TRY(hipSetDevice(m->devices[0]));
            HIP_TRY(hipStreamWaitEvent(m->gather_stream, m->ev_ready, 0));
            HIP_TRY(hipMemcpyAsync(gathered + (size_t)r * block, m->shard[r]->d_pixels, count * sizeof(uint32_t),
                                   hipMemcpyDeviceToDevice, m->gather_stream));
        }
        // ... and a shard's NEXT render must not overwrite its pixel buffer before it has been copied out (ncclSend sits on
        // the shard's own stream and orders that by itself; the copy above sits on the root's gather stream)
        HIP_TRY(hipEventRecord(m->ev_copied, m->gather_stream));
        for (int r = 1; r < m->n; ++r)
            if (m->shard[r]->local_rows > 0) HIP_TRY(hipStreamWaitEvent(m->shard[r]->stream, m->ev_copied, 0));
    }
    HIP_TRY(hipSetDevice(m->devices[0]));
    rc = rt_deinterleave_rows(m->d_full, gathered, m->w, m->h, m->n, m->tile_rows, m->pad_rows, m->devices[0], m->gather_stream);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipEventRecord(m->ev_gathered[slot], m->gather_stream));
    m->slot_used[slot] = true;
    m->frames += 1;
    return RT_OK;
}

// the stream the assembled frame is complete on (readers copy from d_full behind it)
hipStream_t frame_stream(rt_multi *m) { return m->n == 1 ? m->shard[0]->last_stream : m->gather_stream; }

}  // namespace

extern "C" {

RT_API int rt_create_multi_on(rt_ctx **out, int w, int h, const int *devices, int ngpus, int tile_rows) {
    if (!out) return fail(RT_ERR_ARG, "out is null");
    *out = nullptr;
    if (w <= 0 || h <= 0) return fail(RT_ERR_ARG, "image size %dx%d", w, h);
    if (ngpus < 1 || ngpus > 64 || !devices) return fail(RT_ERR_ARG, "ngpus %d", ngpus);
    if (tile_rows == 0) tile_rows = rt::kTileH;
    if (tile_rows < 0 || tile_rows % rt::kTileH != 0) return fail(RT_ERR_ARG, "tile_rows must be a positive multiple of %d", rt::kTileH);
    // the list is either all distinct (one RCCL rank per device) or one device n times (the one-GPU rehearsal, whose
    // transfers are device-to-device copies): RCCL refuses two ranks on a device, and a mixed list would need both
    int n_same = 0, n_pairs = 0;
    for (int i = 0; i < ngpus; ++i) {
        if (devices[i] < 0) return fail(RT_ERR_ARG, "device %d", devices[i]);
        for (int j = 0; j < i; ++j) {
            n_pairs += 1;
            n_same += devices[j] == devices[i] ? 1 : 0;
        }
    }
    if (n_same != 0 && n_same != n_pairs)
        return fail(RT_ERR_ARG, "the device list must name %d different devices, or one device %d times (the one-GPU rehearsal); a mixed list is refused", ngpus, ngpus);
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(RT_ERR_NO_DEVICE, "no HIP device (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    bool repeated = false;
    for (int i = 0; i < ngpus; ++i) {
        if (devices[i] < 0 || devices[i] >= n_dev) return fail(RT_ERR_ARG, "device %d of %d", devices[i], n_dev);
        for (int j = 0; j < i; ++j) repeated = repeated || devices[j] == devices[i];
    }

    rt_ctx *front = new (std::nothrow) rt_ctx();
    rt_multi *m = new (std::nothrow) rt_multi();
    if (!front || !m) {
        delete front;
        delete m;
        return fail(RT_ERR_ALLOC, "host allocation failed");
    }
    front->multi = m;
    front->w = w;
    front->h = h;
    front->device = devices[0];
    front->tile_rows = tile_rows;
    front->local_rows = h;              // the whole image
    m->n = ngpus;
    m->w = w;
    m->h = h;
    m->tile_rows = tile_rows;
    m->emulated = repeated && !g_repeated_counts_as_distinct;
    m->devices.assign(devices, devices + ngpus);
    for (int r = 0; r < ngpus; ++r) m->pad_rows = rows_of(h, r, ngpus, tile_rows) > m->pad_rows ? rows_of(h, r, ngpus, tile_rows) : m->pad_rows;

    auto build = [&]() -> int {
        if (!m->emulated && ngpus > 1) {               // (one device: nothing to move, no communicator, RCCL is not even loaded)
            int rc = load_rccl();
            if (rc != RT_OK) return rc;
            m->comm.assign(ngpus, nullptr);
            RCCL_TRY(g_rccl.CommInitAll(m->comm.data(), ngpus, m->devices.data()));
        }
        for (int r = 0; r < ngpus; ++r) {
            rt_ctx *s = nullptr;
            int rc = rt_create_sharded(&s, w, h, devices[r], r, ngpus, tile_rows);
            if (rc != RT_OK) return rc;
            // ONE decision per context between the hierarchy and the plain sweep: the first shard measures, the others follow
            if (r > 0) s->choice_leader = m->shard[0];
            m->shard.push_back(s);
        }
        HIP_TRY(hipSetDevice(devices[0]));
        HIP_TRY(hipMalloc(&m->d_full, ((size_t)w * h + 4) * sizeof(uint32_t)));
        HIP_TRY(hipMemset(m->d_full, 0, (size_t)w * h * sizeof(uint32_t)));
        if (ngpus == 1)                                 // the one shard's rows are the frame: no slots, no gather, no de-interleave
            return rt_set_pixel_buffer(m->shard[0], m->d_full, (size_t)w * h);
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(hipMalloc(&m->d_gathered[k], ((size_t)ngpus * m->pad_rows * w + 4) * sizeof(uint32_t)));
            HIP_TRY(hipMemset(m->d_gathered[k], 0, (size_t)ngpus * m->pad_rows * w * sizeof(uint32_t)));
            HIP_TRY(hipEventCreateWithFlags(&m->ev_gathered[k], hipEventDisableTiming));
        }
        HIP_TRY(hipStreamCreateWithFlags(&m->gather_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_rendered, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_ready, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_copied, hipEventDisableTiming));
        // the root renders its own rows straight into the receive slot of the frame (begin_frame sets it per frame)
        return rt_set_pixel_buffer(m->shard[0], m->d_gathered[0], (size_t)m->pad_rows * w);
    };
    int rc = build();
    if (rc != RT_OK) {
        char keep[512];
        snprintf(keep, sizeof keep, "%s", rt_last_error());
        rt_destroy(front);
        fail(rc, "%s", keep);
        return rc;
    }
    *out = front;
    return RT_OK;
}

RT_API int rt_create_multi(rt_ctx **out, int w, int h, int ngpus) {
    if (ngpus < 1 || ngpus > 64) return fail(RT_ERR_ARG, "ngpus %d", ngpus);
    int devices[64];
    for (int i = 0; i < ngpus; ++i) devices[i] = i;
    return rt_create_multi_on(out, w, h, devices, ngpus, rt::kTileH);
}

}  // extern "C"

namespace rt {

// Has everything queued on `s` (a stream of `device`) completed?  Asked without blocking, for at most `budget_ms` in all.
static bool stream_drains(int device, hipStream_t s, double *budget_ms) {
    if (!s) return true;
    if (hipSetDevice(device) != hipSuccess) return false;
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return true;
        if (e != hipErrorNotReady) {
            (void)hipGetLastError();
            return false;
        }
        if (*budget_ms <= 0.0) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(500));
        *budget_ms -= 0.5;
    }
}

void multi_destroy(rt_ctx *front) {
    rt_multi *m = front->multi;
    if (!m) return;
    // A broken context (mark_broken) MAY hold a transfer that never completes on the gather stream or on a shard's -- hipFree,
    // hipHostFree and hipStreamDestroy synchronise with such work and would hang with it.  Whether it does is asked, not assumed:
    // its streams are polled (hipStreamQuery, never a blocking wait) for up to two seconds.  When they have all drained -- the
    // failure came before anything was posted, or ncclCommAbort tore the transfers down -- teardown is the ordinary one and
    // nothing is lost; only a context whose streams are still busy after that keeps (leaks) its device memory, page-locked
    // staging, events and streams, and the caller's output buffer stays registered: its teardown returns, it does not tidy up.
    bool stuck = false;
    if (m->broken) {
        double budget_ms = 2000.0;
        for (size_t r = 0; r < m->shard.size() && !stuck; ++r) {
            rt_ctx *s = m->shard[r];
            stuck = !stream_drains(s->device, s->stream, &budget_ms) ||
                    (s->last_stream && s->last_stream != s->stream && !stream_drains(s->device, s->last_stream, &budget_ms));
        }
        if (!stuck && m->gather_stream && !m->devices.empty()) stuck = !stream_drains(m->devices[0], m->gather_stream, &budget_ms);
        if (!stuck)
            for (rt_ctx *s : m->shard) s->abandon_streams = false;      // (rt_destroy of a shard then frees as ever)
    }
    if (m->gather_stream && !m->broken && !m->devices.empty() && hipSetDevice(m->devices[0]) == hipSuccess)
        (void)hipStreamSynchronize(m->gather_stream);   // (it reads the shards' buffers: before they go)
    for (rt_ctx *s : m->shard) rt_destroy(s);           // waits for each shard's stream (a stuck context's shards: neither wait nor free, rt_api.hip)
    if (!stuck && !m->devices.empty() && hipSetDevice(m->devices[0]) == hipSuccess) {
        if (m->pinned_out) (void)hipHostUnregister(m->pinned_out);
        (void)hipFree(m->d_gathered[0]);
        (void)hipFree(m->d_gathered[1]);
        (void)hipFree(m->d_full);
        if (m->ev_ready) (void)hipEventDestroy(m->ev_ready);
        if (m->ev_copied) (void)hipEventDestroy(m->ev_copied);
        if (m->ev_rendered) (void)hipEventDestroy(m->ev_rendered);
        for (hipEvent_t e : m->ev_gathered)
            if (e) (void)hipEventDestroy(e);
        if (m->gather_stream) (void)hipStreamDestroy(m->gather_stream);
    }
    // (mark_broken has aborted and cleared the communicators where the library has ncclCommAbort; where it has not, a
    // communicator that sits in a failed group is not handed to ncclCommDestroy, which may wait for that group: leaked)
    for (rcclComm c : m->comm)
        if (c && !m->broken) (void)g_rccl.CommDestroy(c);
    delete m;
    front->multi = nullptr;
}

int multi_shards(const rt_ctx *front) { return front->multi->n; }
// host waits for the last gathered frame to be assembled (rt_throttle(ctx, 0) on a multi-device context)
int multi_wait_frame(rt_ctx *front) {
    rt_multi *m = front->multi;
    if (m->n == 1 || !m->gather_stream || m->broken) return RT_OK;
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipStreamSynchronize(m->gather_stream));
    return RT_OK;
}
rt_ctx *multi_first_shard(rt_ctx *front) { return front->multi->shard.empty() ? front : front->multi->shard[0]; }
rt_ctx *multi_shard(rt_ctx *front, int r) { return front->multi->shard[(size_t)r]; }
const char *multi_last_kernel(const rt_ctx *front) { return front->multi->shard.empty() ? "" : front->multi->shard[0]->last_kernel; }

#define EACH_SHARD(expr)                          \
    do {                                          \
        {                                         \
            int rb_ = refuse_if_broken(front->multi); \
            if (rb_ != RT_OK) return rb_;         \
        }                                         \
        for (rt_ctx * s : front->multi->shard) {  \
            int rc_ = (expr);                     \
            if (rc_ != RT_OK) return rc_;         \
        }                                         \
        return RT_OK;                             \
    } while (0)

int multi_set_scene(rt_ctx *front, const rt_sphere *spheres, uint32_t count) { EACH_SHARD(rt_set_scene(s, spheres, count)); }
int multi_update_spheres(rt_ctx *front, uint32_t first, uint32_t count, const rt_sphere *spheres) {
    EACH_SHARD(rt_update_spheres_async(s, first, count, spheres, s->stream));
}
int multi_set_camera(rt_ctx *front, const rt_camera *cam) { EACH_SHARD(rt_set_camera(s, cam)); }
int multi_set_mode(rt_ctx *front, int mode) { EACH_SHARD(rt_set_mode(s, mode)); }
int multi_set_pixel_write(rt_ctx *front, int enable) { EACH_SHARD(rt_set_pixel_write(s, enable)); }
int multi_debug_each(rt_ctx *front, int (*fn)(rt_ctx *, int), int arg) { EACH_SHARD(fn(s, arg)); }
// diagnostics: put the context into the state a failed ncclGroupEnd leaves it in
int multi_debug_break(rt_ctx *front) { return mark_broken(front->multi, RT_ERR_HIP, "ncclGroupEnd (injected by rt_debug_break_gather)", "unhandled system error"); }

int multi_reset(rt_ctx *front, bool async) {
    rt_multi *m = front->multi;
    m->launches = 0;
    m->last_ms = 0.0;
    front->current_sample = 0;
    EACH_SHARD(async ? rt_reset_async(s, s->stream) : rt_reset(s));
}

// rt_stream() of a multi-device context: the stream the ASSEMBLED frame is complete on (n > 1: the root's gather stream, behind
// the receives and the de-interleave of the last frame queued; n = 1: the one shard's render stream).  A consumer that reads
// rt_device_pixels behind it sees whole frames, and the next frame's assembly is ordered behind that read in turn.
void *multi_stream(rt_ctx *front) {
    rt_multi *m = front->multi;
    return (void *)(m->n == 1 ? m->shard[0]->stream : m->gather_stream);     // (every shard renders on its own stream: never a caller's, never null)
}

#if RT_DIAGNOSTICS
// rt_debug_set_rccl_library: forget the bound library (contexts created before keep the communicators of the old one: the
// tests create theirs afterwards) and bind `path` at the next multi-device context
int multi_debug_set_rccl(const char *path, int repeated_counts_as_distinct) {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    g_rccl = Rccl{};
    g_rccl_override = path ? path : "";
    g_repeated_counts_as_distinct = repeated_counts_as_distinct != 0;
    return RT_OK;
}
#endif

int multi_render(rt_ctx *front, uint32_t *out_host, int n_samples, bool blocking) {
    rt_multi *m = front->multi;
    if (n_samples < 0) return fail(RT_ERR_ARG, "n_samples < 0");
    {
        int rb = refuse_if_broken(m);
        if (rb != RT_OK) return rb;
    }
    const bool frame_wanted = m->shard[0]->pixel_write != 0 && n_samples > 0;
    if (frame_wanted) {
        int rc = begin_frame(m);
        if (rc != RT_OK) return rc;
    }
    // every device starts its rows before any is waited for
    for (int r = 0; r < m->n; ++r) {
        rt_ctx *s = m->shard[r];
        HIP_TRY(hipSetDevice(s->device));
        if (blocking) HIP_TRY(hipEventRecord(s->ev0, s->stream));
        // (never the blocking probe sequence: the shard that measures hierarchy against sweep would wait for its verdict inside
        // this loop, before the other devices have been given their rows; a measured scene's probes are then whole launches --
        // this frame and the next three -- as on the asynchronous path)
        int rc = rt::render_shard(s, n_samples, false);
        if (rc != RT_OK) return rc;
        if (blocking) HIP_TRY(hipEventRecord(s->ev1, s->stream));
    }
    front->current_sample = m->shard[0]->current_sample;
    if (n_samples > 0) m->launches += 1;
    if (frame_wanted) {
        int rc = gather_and_assemble(m);
        if (rc != RT_OK) return rc;
    }
    if (!blocking) return RT_OK;
    HIP_TRY(hipSetDevice(m->devices[0]));
    if (out_host)
        HIP_TRY(hipMemcpyAsync(out_host, m->d_full, (size_t)m->w * m->h * sizeof(uint32_t), hipMemcpyDeviceToHost, frame_stream(m)));
    HIP_TRY(hipStreamSynchronize(frame_stream(m)));
    double worst = 0.0;
    for (int r = 0; r < m->n; ++r) {
        rt_ctx *s = m->shard[r];
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipStreamSynchronize(s->stream));
        float ms = 0.f;
        if (n_samples > 0 && s->local_rows > 0) HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
        s->last_ms = ms;
        worst = ms > worst ? ms : worst;
    }
    m->last_ms = worst;                 // the frame's kernel time = its slowest shard
    return RT_OK;
}

int multi_read_pixels(rt_ctx *front, uint32_t *out_host) {
    rt_multi *m = front->multi;
    {
        int rb = refuse_if_broken(m);
        if (rb != RT_OK) return rb;
    }
    // shards whose last launches skipped the pixel store pack their rows now; then the usual gather
    bool stale = false;
    for (rt_ctx *s : m->shard) stale = stale || (!s->pixels_current && s->current_sample > 0);
    if (stale) {
        int rb = begin_frame(m);                        // (the root packs into the receive slot of the frame about to be gathered)
        if (rb != RT_OK) return rb;
        std::vector<uint32_t> tmp;
        for (int r = 0; r < m->n; ++r) {
            rt_ctx *s = m->shard[r];
            if (s->local_rows == 0) continue;
            // rt_read_pixels packs into the shard's pixel buffer (the root's is its receive slot); the host copy is a by-product
            tmp.resize((size_t)s->local_rows * m->w);
            int rc = rt_read_pixels(s, tmp.data());
            if (rc != RT_OK) return rc;
        }
        int rc = gather_and_assemble(m);
        if (rc != RT_OK) return rc;
    }
    for (int r = 1; r < m->n; ++r) {                    // async renders on the other devices
        HIP_TRY(hipSetDevice(m->devices[r]));
        HIP_TRY(hipStreamSynchronize(m->shard[r]->stream));
    }
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipMemcpyAsync(out_host, m->d_full, (size_t)m->w * m->h * sizeof(uint32_t), hipMemcpyDeviceToHost, frame_stream(m)));
    HIP_TRY(hipStreamSynchronize(frame_stream(m)));
    return RT_OK;
}

// colour plane / seeds: every shard holds full-size buffers in which only its own rows are live
static void rows_to(const rt_multi *m, int r, const void *src, void *dst, size_t bytes_per_px, bool flipped) {
    const int n_tiles = (m->h + m->tile_rows - 1) / m->tile_rows;
    const size_t row_bytes = (size_t)m->w * bytes_per_px;
    for (int t = r; t < n_tiles; t += m->n)
        for (int y = t * m->tile_rows; y < (t + 1) * m->tile_rows && y < m->h; ++y) {
            const size_t row = flipped ? (size_t)(m->h - 1 - y) : (size_t)y;     // .cl:579: the colour plane is y-flipped
            memcpy(static_cast<char *>(dst) + row * row_bytes, static_cast<const char *>(src) + row * row_bytes, row_bytes);
        }
}

int multi_read_colors(rt_ctx *front, float *out_host) {
    rt_multi *m = front->multi;
    std::vector<float> tmp((size_t)3 * m->w * m->h);
    for (int r = 0; r < m->n; ++r) {
        int rc = rt_read_colors(m->shard[r], tmp.data());
        if (rc != RT_OK) return rc;
        rows_to(m, r, tmp.data(), out_host, 12, true);
    }
    return RT_OK;
}

int multi_read_seeds(rt_ctx *front, uint32_t *out_host) {
    rt_multi *m = front->multi;
    std::vector<uint32_t> tmp((size_t)2 * m->w * m->h);
    for (int r = 0; r < m->n; ++r) {
        int rc = rt_read_seeds(m->shard[r], tmp.data());
        if (rc != RT_OK) return rc;
        rows_to(m, r, tmp.data(), out_host, 8, false);
    }
    return RT_OK;
}

int multi_get_stats(rt_ctx *front, rt_stats *out) {
    rt_multi *m = front->multi;
    rt_stats sum{};
    for (rt_ctx *s : m->shard) {
        rt_stats st;
        int rc = rt_get_stats(s, &st);
        if (rc != RT_OK) return rc;
        sum.samples += st.samples;
        sum.closest_rays += st.closest_rays;
        sum.shadow_rays += st.shadow_rays;
        sum.sphere_tests += st.sphere_tests;
        sum.rng_draws += st.rng_draws;
    }
    sum.launches = m->launches;
    sum.last_kernel_ms = m->last_ms;
    *out = sum;
    return RT_OK;
}

int multi_device_pixels(rt_ctx *front, void **dptr, size_t *count) {
    *dptr = front->multi->d_full;
    *count = (size_t)front->multi->w * (size_t)front->multi->h;
    return RT_OK;
}

int multi_pin_output(rt_ctx *front, uint32_t *out_host, size_t count) {
    rt_multi *m = front->multi;
    HIP_TRY(hipSetDevice(m->devices[0]));
    if (m->pinned_out) {
        HIP_TRY(hipStreamSynchronize(frame_stream(m)));
        (void)hipHostUnregister(m->pinned_out);
        m->pinned_out = nullptr;
    }
    if (!out_host) return RT_OK;
    if (count < (size_t)m->w * (size_t)m->h) return fail(RT_ERR_ARG, "output buffer of %zu < %zu elements", count, (size_t)m->w * (size_t)m->h);
    HIP_TRY(hipHostRegister(out_host, count * sizeof(uint32_t), hipHostRegisterDefault));
    m->pinned_out = out_host;
    return RT_OK;
}

}  // namespace rt
