// stream_order.hip -- does kernel B see what kernel A wrote just before it ON THE SAME STREAM, when
// several processes share the GPU and each drives several non-blocking streams?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/stream_order.hip -o /tmp/stream_order
//   for p in 1 2 3 4; do /tmp/stream_order 6 4000 & done; wait
// Per stream and iteration k: kernel A stores k into every word of a 2 MiB buffer, kernel B (launched
// right after it on the same stream, after a long-running unrelated kernel C) counts the words that
// are not k.  Prints the number of (iteration, stream) pairs with a stale word.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <thread>
#include <vector>
#include <unistd.h>

__global__ void fill(unsigned *buf, size_t n, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = v;
}
__global__ void check(const unsigned *buf, size_t n, unsigned v, unsigned long long *bad) {
    unsigned long long b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b += (buf[i] != v);
    if (b) atomicAdd(bad, b);
}
// a long kernel that stores at its very END (like a render kernel writing its final state)
__global__ void slow_fill(unsigned *buf, size_t n, unsigned v, int iters) {
    float x = threadIdx.x;
    for (int i = 0; i < iters + (int)(blockIdx.x & 7) * iters; ++i) x = x * 1.0000001f + 0.5f;     // uneven block durations
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = v + (x == 12345.f);
}
__global__ void spin(float *out, int iters) {                      // keeps the machine busy like a render
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0000001f + 0.5f;
    if (x == 12345.f) out[0] = x;
}

int main(int argc, char **argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 6, iters = argc > 2 ? atoi(argv[2]) : 2000;
    const int with_wait = argc > 3 ? atoi(argv[3]) : 0;      // 1: a hipStreamWaitEvent on an event of another stream precedes A
    const size_t n = 512 * 1024;
    std::vector<hipStream_t> st(S);
    std::vector<unsigned *> buf(S);
    std::vector<unsigned long long *> bad(S);
    float *sink;
    hipMalloc(&sink, 4);
    for (int s = 0; s < S; ++s) {
        hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
        hipMalloc(&buf[s], n * 4);
        hipMalloc(&bad[s], 8);
        hipMemset(bad[s], 0, 8);
        hipMemset(buf[s], 0, n * 4);
    }
    hipStream_t side;
    hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
    std::vector<hipEvent_t> ev(S);
    for (int s = 0; s < S; ++s) hipEventCreateWithFlags(&ev[s], hipEventDisableTiming);
    // 2: a second host thread that keeps copying device memory to pinned host memory on a pool of its own
    // streams and waiting for it (what a host-staged collective's worker thread does)
    std::atomic<bool> stop{false};
    std::thread worker;
    if (with_wait >= 2) worker = std::thread([&]() {
        hipSetDevice(0);
        hipStream_t ws[4];
        unsigned *d; unsigned *h;
        hipMalloc(&d, 1 << 20); hipHostMalloc(&h, 1 << 20);
        for (auto &w : ws) hipStreamCreateWithFlags(&w, hipStreamNonBlocking);
        hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
        for (int i = 0; !stop.load(); ++i) {
            hipStream_t w = ws[i & 3];
            if (with_wait >= 5) {            // 5: the worker also launches kernels and chains its streams with events
                hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, w, sink, 500);
                hipEventRecord(e, w);
                hipStreamWaitEvent(ws[(i + 1) & 3], e, 0);
            }
            hipMemcpyAsync(h, d, 1 << 20, hipMemcpyDeviceToHost, w);
            hipEventRecord(e, w);
            hipStreamSynchronize(w);
            if ((i & 63) == 0) { void *p; hipHostMalloc(&p, 1 << 16); hipHostFree(p); }
        }
    });
    hipDeviceSynchronize();
    unsigned long long total_events = 0;
    std::vector<unsigned long long> prev(S, 0);
    for (int k = 1; k <= iters; ++k) {
        const int s = k % S;
        hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, st[s], sink, 20000);
        if (with_wait) {                     // what a collective's work.wait() does to the caller's stream
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, side, sink, 2000);
            hipEventRecord(ev[s], side);
            hipStreamWaitEvent(st[s], ev[s], 0);
        }
        if (with_wait >= 3) {                // X: long kernel storing late, then an EVENT RECORD on this very stream, then Y and Z
            hipLaunchKernelGGL(slow_fill, dim3(2048), dim3(256), 0, st[s], buf[s], n, 0x80000000u | (unsigned)k, 3000);
            hipEventRecord(ev[s], st[s]);
            if (with_wait >= 4) hipStreamWaitEvent(side, ev[s], 0);       // and somebody else waits for it
        }
        hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, st[s], buf[s], n, (unsigned)k);
        hipLaunchKernelGGL(check, dim3(256), dim3(256), 0, st[s], buf[s], n, (unsigned)k, bad[s]);
        if (k % (8 * S) == 0) {
            hipDeviceSynchronize();
            for (int q = 0; q < S; ++q) {
                unsigned long long h = 0;
                hipMemcpy(&h, bad[q], 8, hipMemcpyDeviceToHost);
                if (h != prev[q]) { total_events += 1; prev[q] = h; }
            }
        }
    }
    hipDeviceSynchronize();
    stop.store(true);
    if (worker.joinable()) worker.join();
    unsigned long long words = 0;
    for (int q = 0; q < S; ++q) { unsigned long long h = 0; hipMemcpy(&h, bad[q], 8, hipMemcpyDeviceToHost); words += h; }
    printf("pid %d: %d streams, %d iterations: %llu stale words, seen in %llu checks\n", (int)getpid(), S, iters, words, total_events);
    return 0;
}
