// valu_ops3.hip -- third batch: (1) the 16-bit three-operand maxima and friends -- do they sit in the two-cycle class like
// v_max_i16 / v_add_u16? (2) dependent chains: SIMD cycles per wave instruction when a wave's instructions depend on each
// other (1 chain) or not (8 chains), at 1, 2 and 4 waves per SIMD -- what a strip of DP cells (each cell waits for its left
// neighbour) can expect from the two classes at the occupancy nwprofile.hip runs at (2 waves per SIMD).
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -w tools/valu_ops3.hip -o /tmp/valu_ops3 && /tmp/valu_ops3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define OPS(X)                                                                                           \
    X(0, "v_add_u32", "v_add_u32 %0, %0, %1")                                                            \
    X(1, "v_max_i32", "v_max_i32 %0, %0, %1")                                                            \
    X(2, "v_max3_i32", "v_max3_i32 %0, %0, %1, %2")                                                      \
    X(3, "v_max_i16", "v_max_i16 %0, %0, %1")                                                            \
    X(4, "v_max_u16", "v_max_u16 %0, %0, %1")                                                            \
    X(5, "v_min_u16", "v_min_u16 %0, %0, %1")                                                            \
    X(6, "v_max3_i16", "v_max3_i16 %0, %0, %1, %2")                                                      \
    X(7, "v_max3_u16", "v_max3_u16 %0, %0, %1, %2")                                                      \
    X(8, "v_min3_u16", "v_min3_u16 %0, %0, %1, %2")                                                      \
    X(9, "v_add_u16", "v_add_u16 %0, %0, %1")                                                            \
    X(10, "v_sub_u16", "v_sub_u16 %0, %0, %1")                                                           \
    X(11, "v_max_i16 (VOP3 e64)", "v_max_i16_e64 %0, %0, %1")                                            \
    X(12, "v_add_u16 sdwa byte", "v_add_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1") \
    X(13, "v_add_u32 sdwa sext byte", "v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1") \
    X(14, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2")                                                     \
    X(15, "v_mad_u16", "v_mad_u16 %0, %0, %1, %2")                                                       \
    X(16, "v_med3_i16", "v_med3_i16 %0, %0, %1, %2")                                                     \
    X(17, "v_mov_b32 dpp wave_shr", "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")        \
    X(18, "v_lshrrev_b32 16", "v_lshrrev_b32 %0, 16, %0")                                                \
    X(19, "v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %1, %2")                                               \
    X(20, "v_max_f32", "v_max_f32 %0, %0, %1")                                                           \
    X(21, "v_add_f32", "v_add_f32 %0, %0, %1")                                                           \
    X(22, "v_max_f16", "v_max_f16 %0, %0, %1")                                                           \
    X(23, "v_add_f16", "v_add_f16 %0, %0, %1")                                                           \
    X(24, "v_max3_f16", "v_max3_f16 %0, %0, %1, %2")

template <int OP, int CHAINS>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t r[8];
    for (int i = 0; i < 8; ++i) r[i] = seed * (threadIdx.x + 1) + i;
    uint32_t c = seed | 1, d = seed * 7 + 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 64 / CHAINS; ++rep) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
#define X(ID, NAME, TEXT) if constexpr (OP == ID) asm volatile(TEXT : "+v"(r[i]) : "v"(c), "v"(d));
                OPS(X)
#undef X
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int CHAINS>
double run(uint32_t *out, int waves_per_simd) {
    const int iters = 2000, blocks = 256 * 4 * waves_per_simd;   // single-wave workgroups: 4 SIMDs x waves_per_simd per CU
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(64), 0, 0, out, 10, 3u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(64), 0, 0, out, iters, 3u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms * 1e-3 * 2.4e9 / ((double)iters * 64 * waves_per_simd);   // SIMD cycles per wave instruction
}

template <int OP>
void row(const char *name, uint32_t *out) {
    printf("%-26s", name);
    for (int waves : {1, 2, 4}) printf("  %5.2f /%5.2f", run<OP, 1>(out, waves), run<OP, 8>(out, waves));
    printf("\n");
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 4096 * 64 * 4);
    for (int warm = 0; warm < 300; ++warm) hipLaunchKernelGGL((k<0, 8>), dim3(4096), dim3(64), 0, 0, out, 2000, 3u);   // clocks
    hipDeviceSynchronize();
    printf("SIMD cycles per wave instruction (2.4 GHz nominal): dependent chain / 8 independent chains\n");
    printf("%-26s  %13s  %13s  %13s\n", "", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
#define X(ID, NAME, TEXT) row<ID>(NAME, out);
    OPS(X)
#undef X
    return 0;
}
