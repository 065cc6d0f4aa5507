// fetch_size.hip -- what FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) report for a KNOWN byte count in the render kernels' own access shapes.
// bench.py's `roofline.traffic` quotes the factors measured here (profiles/r06_fetch_size_calibration.json) instead of assuming the guide's
// "half-count" for wide streams applies to them.  Every kernel moves exactly the bytes its name says over a 1920x1080 image:
//   read_u2_tile8      one wavefront per 8x8 pixel tile, each lane one uint2 (the seed pair): 64-byte row segments   (rt_trace_*_w1)
//   read_u2_tile32     256-thread workgroups of four 8x8 sub-tiles side by side: 256 contiguous bytes per row       (the 4-wavefront instances)
//   read_f4_stream     16 bytes per lane, fully coalesced                                                             (the guide's calibration shape)
//   write_frame_tile8  the epilogue's stores per pixel: uint2 seeds + three floats of colour + one packed pixel (8 + 12 + 4 bytes)
// Before each measured launch a flush kernel streams 1 GiB so that nothing of the image is left in L2 or the Infinity Cache; launches named
// *_warm run back to back without it (what a frame loop sees).     hipcc --offload-arch=gfx950 -O2 fetch_size.hip -o fetch_size
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int W = 1920, H = 1080;

__global__ void __launch_bounds__(64) read_u2_tile8(const uint2 *img, unsigned *sink) {
    const int x = blockIdx.x * 8 + (threadIdx.x & 7), y = blockIdx.y * 8 + (threadIdx.x >> 3);
    const uint2 v = img[(size_t)y * W + x];
    if ((v.x ^ v.y) == 0x12345678u) atomicAdd(sink, 1u);
}
__global__ void __launch_bounds__(256) read_u2_tile32(const uint2 *img, unsigned *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = blockIdx.x * 32 + wave * 8 + (lane & 7), y = blockIdx.y * 8 + (lane >> 3);
    const uint2 v = img[(size_t)y * W + x];
    if ((v.x ^ v.y) == 0x12345678u) atomicAdd(sink, 1u);
}
__global__ void __launch_bounds__(256) read_f4_stream(const float4 *buf, size_t n, unsigned *sink) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float4 v = buf[i];
        if (v.x + v.y + v.z + v.w == 12345.678f) atomicAdd(sink, 1u);
    }
}
__global__ void __launch_bounds__(64) write_frame_tile8(uint2 *seeds, float *colors, unsigned *pixels) {
    const int x = blockIdx.x * 8 + (threadIdx.x & 7), y = blockIdx.y * 8 + (threadIdx.x >> 3);
    const size_t gid = (size_t)y * W + x, ci = (size_t)(H - 1 - y) * W + x;
    seeds[gid] = make_uint2((unsigned)gid, (unsigned)ci);
    colors[3 * ci] = 0.25f; colors[3 * ci + 1] = 0.5f; colors[3 * ci + 2] = 0.75f;
    pixels[gid] = 0x00808080u;
}
__global__ void __launch_bounds__(256) flush_stream(const float4 *buf, size_t n, unsigned *sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += buf[i].x;
    if (acc == 12345.678f) atomicAdd(sink, 1u);
}

int main() {
    const size_t px = (size_t)W * H;
    uint2 *img, *seeds; float *colors; unsigned *pixels, *sink; float4 *big;
    const size_t big_n = (size_t)1 << 26;       // 1 GiB of float4
    CK(hipMalloc(&img, px * 8)); CK(hipMalloc(&seeds, px * 8)); CK(hipMalloc(&colors, px * 12)); CK(hipMalloc(&pixels, px * 4));
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&big, big_n * 16));
    CK(hipMemset(img, 1, px * 8)); CK(hipMemset(big, 0, big_n * 16)); CK(hipMemset(sink, 0, 4));
    auto flush = [&]() { flush_stream<<<4096, 256>>>(big, big_n, sink); };
    const dim3 g8(W / 8, H / 8), g32(W / 32, H / 8);
    for (int rep = 0; rep < 3; ++rep) {
        flush(); read_u2_tile8<<<g8, 64>>>(img, sink);
        flush(); read_u2_tile32<<<g32, 256>>>(img, sink);
        flush(); read_f4_stream<<<(unsigned)((px / 2 + 255) / 256), 256>>>(reinterpret_cast<const float4 *>(img), px / 2, sink);
        flush(); write_frame_tile8<<<g8, 64>>>(seeds, colors, pixels);
    }
    CK(hipDeviceSynchronize());
    printf("{\"bytes\": {\"read_u2_tile8\": %zu, \"read_u2_tile32\": %zu, \"read_f4_stream\": %zu, \"write_frame_tile8\": %zu, \"flush_stream\": %zu}}\n",
           px * 8, px * 8, px * 8, px * 24, big_n * 16);
    return 0;
}
