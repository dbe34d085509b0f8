// Microbenchmark: LDS cycles per wave-instruction on gfx950 for the access shapes k_short_tiled uses (aligned / unaligned wide
// reads at string-like addresses, ds_or, ds_add_rtn with repeated addresses, byte stores, 128-bit clears).
// Four waves per CU (one per SIMD) issue the same instruction back to back, eight independent ones per iteration, so the
// LDS pipeline -- shared by the CU -- is the limit: cycles per instruction = elapsed CU cycles / (4 waves x instructions).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_ops.hip -o tools/bin/lds_ops && tools/bin/lds_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kIters = 2000, kUnroll = 8;

enum Op { kRead32, kRead64, kRead64U, kRead128, kRead128U, kRead128A16, kOr32, kAddRtn, kWrite128, kWrite8, kReadU8, kRead2x32, kWrite32, kWrite64 };

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, int stride_mode) {
    __shared__ __attribute__((aligned(4096))) unsigned char lds[65536 - 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (65536 - 1024) / 4; i += 256) ((unsigned *)lds)[i] = i * 2654435761u;
    __syncthreads();
    unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds + wave * 12288;
    unsigned addr;
    if (OP == kRead32 || OP == kOr32 || OP == kWrite32) addr = base + (lane & 31) * 4 + (lane >> 5) * 0;   // [entry][lane&31] dwords shared by l / l+32
    else if (OP == kRead64 || OP == kWrite64) addr = base + lane * 8;
    else if (OP == kRead128 || OP == kWrite128) addr = base + lane * 16;
    else if (OP == kRead128A16) addr = base + ((lane * 7) & ~15);          // aligned, overlapping strings
    else if (OP == kAddRtn) addr = base + ((lane * 7) % 20) * 4;           // ~20 distinct counters
    else if (OP == kWrite8) addr = base + ((lane * 37) & 1023);
    else addr = base + (stride_mode >> 8 ? (stride_mode >> 8) : 1) + lane * ((stride_mode & 255) ? (stride_mode & 255) : 6);   // string-like: offset | stride
    unsigned acc = 0;
    unsigned v0 = lane, v1 = 0, v2 = 0, v3 = 0;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if constexpr (OP == kRead32) { unsigned r; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(128 * u)); acc ^= r; }
            else if constexpr (OP == kReadU8) { unsigned r; asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(u)); acc ^= r; }
            else if constexpr (OP == kRead2x32) { unsigned long long r; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(addr & ~3u), "n"(2 * u), "n"(2 * u + 1)); acc ^= (unsigned)r ^ (unsigned)(r >> 32); }
            else if constexpr (OP == kRead64 || OP == kRead64U) { unsigned long long r; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(512 * u)); acc ^= (unsigned)r ^ (unsigned)(r >> 32); }
            else if constexpr (OP == kRead128 || OP == kRead128U || OP == kRead128A16) { typedef unsigned v4u __attribute__((ext_vector_type(4))); v4u r; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(1024 * u)); acc ^= r.x ^ r.y ^ r.z ^ r.w; }
            else if constexpr (OP == kOr32) { asm volatile("ds_or_b32 %0, %1 offset:%2" :: "v"(addr), "v"(v0), "n"(128 * u) : "memory"); }
            else if constexpr (OP == kWrite32) { asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(v0), "n"(128 * u) : "memory"); }
            else if constexpr (OP == kWrite64) { unsigned long long w = v0; asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(w), "n"(512 * u) : "memory"); }
            else if constexpr (OP == kAddRtn) { unsigned r; asm volatile("ds_add_rtn_u32 %0, %1, %2 offset:%3" : "=v"(r) : "v"(addr), "v"(v0), "n"(128 * u) : "memory"); acc ^= r; }
            else if constexpr (OP == kWrite128) { typedef unsigned v4u __attribute__((ext_vector_type(4))); v4u z = {v0, v1, v2, v3}; asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(z), "n"(1024 * u) : "memory"); }
            else if constexpr (OP == kWrite8) { asm volatile("ds_write_b8 %0, %1 offset:%2" :: "v"(addr), "v"(v0), "n"(u) : "memory"); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int OP> static void run(const char *name, int stride_mode = 0) {
    unsigned *out; (void)hipMalloc(&out, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, out, stride_mode);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, out, stride_mode);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    int clock_khz = 0; (void)hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0);
    const double cycles = ms * 1e-3 * clock_khz * 1e3;
    printf("%-44s %8.1f us  %6.2f CU cycles per wave-instruction (4 waves per CU)\n", name, ms * 1e3, cycles / (4.0 * kIters * kUnroll));
    (void)hipFree(out);
}

int main() {
    run<kRead32>("ds_read_b32 [entry][lane&31]");
    run<kRead2x32>("ds_read2_b32 aligned, string-like");
    run<kReadU8>("ds_read_u8 string-like (6 B apart)");
    run<kRead64>("ds_read_b64 aligned, 8 B apart");
    run<kRead64U>("ds_read_b64 UNALIGNED, 6 B apart");
    run<kRead64U>("ds_read_b64 UNALIGNED, 13 B apart", 13);
    run<kRead128>("ds_read_b128 aligned, 16 B apart");
    run<kRead128A16>("ds_read_b128 aligned, overlapping");
    run<kRead128U>("ds_read_b128 UNALIGNED, 6 B apart");
    run<kRead128U>("ds_read_b128 UNALIGNED, 13 B apart", 13);
    run<kRead64U>("ds_read_b64 dword-aligned (+4), 12 B apart", (4 << 8) | 12);
    run<kRead128U>("ds_read_b128 dword-aligned (+4), 12 B apart", (4 << 8) | 12);
    run<kRead128U>("ds_read_b128 8-aligned (+8), 24 B apart", (8 << 8) | 24);
    run<kRead64U>("ds_read_b64 2-aligned (+2), 6 B apart", (2 << 8) | 6);
    run<kOr32>("ds_or_b32 [entry][lane&31]");
    run<kWrite32>("ds_write_b32 [entry][lane&31]");
    run<kWrite64>("ds_write_b64 8 B apart");
    run<kAddRtn>("ds_add_rtn_u32, ~20 distinct words");
    run<kWrite128>("ds_write_b128 16 B apart");
    run<kWrite8>("ds_write_b8 scattered");
    return 0;
}
