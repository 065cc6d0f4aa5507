// rccl_double.cpp -- a test double for the slice of RCCL that rt_multi.hip binds (test infrastructure: built by
// tests/test_gpu_features.py with hipcc into a scratch directory, bound through rt_debug_set_rccl_library of the
// diagnostics library; never part of the product).
//
// It keeps the semantics the library relies on: inside ncclGroupStart .. ncclGroupEnd, ncclSend(to peer p) on rank r's
// communicator and ncclRecv(from peer r) on rank p's pair up; the transfer runs once BOTH streams have reached it, the
// receiver's stream continues when the data has landed, the sender's when it has left -- here one event each way around a
// device-to-device copy on the receiving stream.  An unmatched or mismatched operation fails ncclGroupEnd.  And it can be
// told to fail: rccl_double_fail("ncclSend", 2) makes the second ncclSend from now on return ncclUnhandledCudaError.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

namespace {
struct Comm { int rank, nranks, device; bool aborted; };
struct Op { bool send; void *buf; size_t count; int type, peer; Comm *comm; hipStream_t stream; };
std::vector<Op> g_ops;
int g_depth = 0;
const char *const kNames[7] = { "ncclCommInitAll", "ncclGroupStart", "ncclSend", "ncclRecv", "ncclGroupEnd", "ncclCommAbort", "ncclCommDestroy" };
long g_calls[7] = { 0, 0, 0, 0, 0, 0, 0 }, g_fail_at[7] = { 0, 0, 0, 0, 0, 0, 0 };
long g_copies = 0;
size_t bytes_of(int type) { return (type == 0 || type == 1) ? 1 : ((type == 4 || type == 5 || type == 8) ? 8 : 4); }   // ncclInt8/Uint8, Int64/Uint64/Float64
int count_call(int k) {            // 0 = go on; otherwise the error this call has been told to return
    g_calls[k] += 1;
    return (g_fail_at[k] != 0 && g_calls[k] == g_fail_at[k]) ? 1 /* ncclUnhandledCudaError */ : 0;
}
}  // namespace

extern "C" {

__attribute__((visibility("default"))) void rccl_double_reset() {
    for (int k = 0; k < 7; ++k) g_calls[k] = g_fail_at[k] = 0;
    g_copies = 0;
    g_ops.clear();
    g_depth = 0;
}
// the `kth` call of `function` counted from now (1 = the next one) fails; 0 = never
__attribute__((visibility("default"))) int rccl_double_fail(const char *function, long kth) {
    for (int k = 0; k < 7; ++k)
        if (!strcmp(function, kNames[k])) {
            g_fail_at[k] = kth ? g_calls[k] + kth : 0;
            return 0;
        }
    return -1;
}
// out[0..6] = calls of the seven functions in kNames order, out[7] = device-to-device copies queued
__attribute__((visibility("default"))) void rccl_double_counts(long *out) {
    for (int k = 0; k < 7; ++k) out[k] = g_calls[k];
    out[7] = g_copies;
}

__attribute__((visibility("default"))) int ncclCommInitAll(void **comms, int n, const int *devices) {
    if (int e = count_call(0)) return e;
    if (!comms || n < 1) return 4;                          // ncclInvalidArgument
    for (int r = 0; r < n; ++r) comms[r] = new Comm{ r, n, devices ? devices[r] : r, false };
    return 0;
}
__attribute__((visibility("default"))) int ncclCommDestroy(void *c) {
    count_call(6);
    delete static_cast<Comm *>(c);
    return 0;
}
__attribute__((visibility("default"))) int ncclCommAbort(void *c) {
    count_call(5);
    delete static_cast<Comm *>(c);
    return 0;
}
__attribute__((visibility("default"))) int ncclGroupStart() {
    if (int e = count_call(1)) return e;
    g_depth += 1;
    return 0;
}
static int post(bool send, void *buf, size_t count, int type, int peer, void *comm, hipStream_t stream) {
    if (int e = count_call(send ? 2 : 3)) return e;
    Comm *c = static_cast<Comm *>(comm);
    if (!c || !buf || peer < 0 || peer >= c->nranks || peer == c->rank) return 4;
    if (g_depth == 0) return 5;                             // ncclInvalidUsage: this double pairs operations at ncclGroupEnd only
    g_ops.push_back(Op{ send, buf, count, type, peer, c, stream });
    return 0;
}
__attribute__((visibility("default"))) int ncclSend(const void *buf, size_t count, int type, int peer, void *comm, hipStream_t stream) {
    return post(true, const_cast<void *>(buf), count, type, peer, comm, stream);
}
__attribute__((visibility("default"))) int ncclRecv(void *buf, size_t count, int type, int peer, void *comm, hipStream_t stream) {
    return post(false, buf, count, type, peer, comm, stream);
}
__attribute__((visibility("default"))) int ncclGroupEnd() {
    const int told = count_call(4);
    if (g_depth == 0) return 5;
    if (--g_depth > 0) return told;
    std::vector<Op> ops;
    ops.swap(g_ops);
    if (told) return told;                                  // the group is dropped: nothing of it is queued
    std::vector<bool> used(ops.size(), false);
    int bad = 0;
    for (size_t i = 0; i < ops.size(); ++i) {
        if (ops[i].send) continue;
        const Op &rv = ops[i];
        size_t j = 0;
        for (; j < ops.size(); ++j)
            if (!used[j] && ops[j].send && ops[j].comm->rank == rv.peer && ops[j].peer == rv.comm->rank) break;
        if (j == ops.size() || ops[j].count != rv.count || ops[j].type != rv.type) { bad = 5; continue; }
        used[i] = used[j] = true;
        const Op &sn = ops[j];
        hipEvent_t sent = nullptr, landed = nullptr;
        if (hipSetDevice(sn.comm->device) != hipSuccess || hipEventCreateWithFlags(&sent, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(sent, sn.stream) != hipSuccess) return 1;
        if (hipSetDevice(rv.comm->device) != hipSuccess || hipEventCreateWithFlags(&landed, hipEventDisableTiming) != hipSuccess ||
            hipStreamWaitEvent(rv.stream, sent, 0) != hipSuccess ||
            hipMemcpyAsync(rv.buf, sn.buf, rv.count * bytes_of(rv.type), hipMemcpyDeviceToDevice, rv.stream) != hipSuccess ||
            hipEventRecord(landed, rv.stream) != hipSuccess) return 1;
        if (hipSetDevice(sn.comm->device) != hipSuccess || hipStreamWaitEvent(sn.stream, landed, 0) != hipSuccess) return 1;
        (void)hipEventDestroy(sent);                        // (released by the runtime once they have completed)
        (void)hipEventDestroy(landed);
        g_copies += 1;
    }
    for (size_t i = 0; i < ops.size(); ++i)
        if (!used[i]) bad = 5;                              // a send nobody receives, a receive nobody sends
    return bad;
}
__attribute__((visibility("default"))) const char *ncclGetErrorString(int e) {
    switch (e) {
        case 0: return "no error";
        case 1: return "unhandled cuda error (rccl_double)";
        case 4: return "invalid argument (rccl_double)";
        case 5: return "invalid usage (rccl_double)";
        default: return "unknown result code (rccl_double)";
    }
}

}  // extern "C"
