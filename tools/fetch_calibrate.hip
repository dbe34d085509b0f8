// Calibrates rocprofv3's FETCH_SIZE on gfx950 for the access patterns of this library (MI355X_MICROARCH.md, "HBM": the counter
// reports half the bytes of a wide coalesced stream; "other access widths ... calibrate on a known byte count in your own
// access pattern"). 1 GiB per kernel -- past the 256 MiB Infinity Cache -- read once with 16-byte loads:
//   k_stream        lane i reads bytes [16 i, 16 i + 16): every byte, coalesced (the tile and segment copies)
//   k_sector_stride lane i reads 16 bytes at 64 i: one window in every 64-byte sector, a wave covers 32 lines (string windows
//                   of consecutive ~64-byte pairs: step B of the tiled kernel)
//   k_line_stride   lane i reads 16 bytes at 128 i: one window per 128-byte line
//   k_scattered     as k_sector_stride, but each wave's windows lie in 64 different, pseudo-randomly chosen lines (the sorted work
//                   items of the bit-parallel kernels)
// build: hipcc --offload-arch=gfx950 -O2 tools/fetch_calibrate.hip -o tools/bin/fetch_calibrate
// run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o cal -- tools/bin/fetch_calibrate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

using u4 = __attribute__((ext_vector_type(4))) uint32_t;
constexpr uint64_t kBytes = 1ull << 30;

__global__ void k_stream(const u4 *in, uint32_t *sink) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u4 v = in[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) *sink = 1;
}
__global__ void k_sector_stride(const u4 *in, uint32_t *sink) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u4 v = in[i * 4 + 1];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) *sink = 1;
}
__global__ void k_line_stride(const u4 *in, uint32_t *sink) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u4 v = in[i * 8 + 1];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) *sink = 1;
}
__global__ void k_scattered(const u4 *in, uint32_t *sink) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t sectors = kBytes / 64;                       // a power of two: an odd multiplier permutes the sectors
    const uint64_t s = (i * 0x9E3779B97F4A7C15ull + 12345u) & (sectors - 1);
    const u4 v = in[s * 4 + 1];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) *sink = 1;
}

int main() {
    u4 *buf = nullptr;
    uint32_t *sink = nullptr;
    if (hipMalloc((void **)&buf, kBytes) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess) { printf("allocation failed\n"); return 1; }
    (void)hipMemset(buf, 1, kBytes);
    (void)hipMemset(sink, 0, 4);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto timed = [&](const char *name, auto kernel, uint64_t threads, uint64_t touched) {
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kernel, dim3((uint32_t)(threads / 256)), dim3(256), 0, 0, buf, sink);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%-16s %6.1f MiB in 16-byte windows, %7.1f MiB of 64-byte sectors touched, %.3f ms\n", name, threads * 16 / 1048576.0, touched / 1048576.0, ms);
        }
    };
    timed("k_stream", k_stream, kBytes / 16, kBytes);
    timed("k_sector_stride", k_sector_stride, kBytes / 64, kBytes);
    timed("k_line_stride", k_line_stride, kBytes / 128, kBytes / 2);
    timed("k_scattered", k_scattered, kBytes / 64, kBytes);
    return 0;
}
