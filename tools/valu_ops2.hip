// valu_ops2.hip -- SIMD cycles per wave instruction, second batch: which VALU operations belong to the two-cycle class
// (v_add_u32, v_and / v_or / v_xor, three-register v_bitop3: profiles/r1/valu_ops_issue_rates.txt) and which to the
// four-cycle class -- integer max / min, float add / max / fma, conversions, packed 16- and 32-bit forms, gfx950's
// three-operand maxima. 8 independent chains per wave, 4 waves per SIMD: issue-bound, not latency-bound.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -w tools/valu_ops2.hip -o /tmp/valu_ops2 && /tmp/valu_ops2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define OPS(X)                                                                                           \
    X(0, "v_add_u32", "v_add_u32 %0, %0, %1")                                                            \
    X(1, "v_sub_u32", "v_sub_u32 %0, %0, %1")                                                            \
    X(2, "v_max_u32", "v_max_u32 %0, %0, %1")                                                            \
    X(3, "v_max_i32", "v_max_i32 %0, %0, %1")                                                            \
    X(4, "v_max_f32", "v_max_f32 %0, %0, %1")                                                            \
    X(5, "v_add_f32", "v_add_f32 %0, %0, %1")                                                            \
    X(6, "v_max3_f32", "v_max3_f32 %0, %0, %1, %2")                                                      \
    X(7, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2")                                                        \
    X(8, "v_fma_mix_f32 (f16 lo)", "v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]")                   \
    X(9, "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 %0, %0")                                                  \
    X(10, "v_pk_add_f32", "v_pk_add_f32 %3, %3, %4")                                                     \
    X(11, "v_pk_fma_f32", "v_pk_fma_f32 %3, %3, %4, %4")                                                 \
    X(12, "v_pk_max_f16", "v_pk_max_f16 %0, %0, %1")                                                     \
    X(13, "v_pk_add_f16", "v_pk_add_f16 %0, %0, %1")                                                     \
    X(14, "v_pk_maximum3_f16", "v_pk_maximum3_f16 %0, %0, %1, %2")                                       \
    X(15, "v_maximum3_f32", "v_maximum3_f32 %0, %0, %1, %2")                                             \
    X(16, "v_med3_i32", "v_med3_i32 %0, %0, %1, %2")                                                     \
    X(17, "v_min3_u32", "v_min3_u32 %0, %0, %1, %2")                                                     \
    X(18, "v_sad_u8", "v_sad_u8 %0, %0, %1, %2")                                                         \
    X(19, "v_mov_b32", "v_mov_b32 %0, %1")                                                               \
    X(20, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1")                                              \
    X(21, "v_add_lshl_u32", "v_add_lshl_u32 %0, %0, %1, 1")                                              \
    X(22, "v_xad_u32", "v_xad_u32 %0, %0, %1, %2")                                                       \
    X(23, "v_not_b32", "v_not_b32 %0, %0")                                                               \
    X(24, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %2")                                                       \
    X(25, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1")                                                   \
    X(26, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1")                                                     \
    X(27, "v_max_i16", "v_max_i16 %0, %0, %1")                                                           \
    X(28, "v_add_u16", "v_add_u16 %0, %0, %1")                                                           \
    X(29, "v_pk_min_u16", "v_pk_min_u16 %0, %0, %1")                                                     \
    X(30, "v_pk_sub_i16", "v_pk_sub_i16 %0, %0, %1")                                                     \
    X(31, "v_pk_lshlrev_b16", "v_pk_lshlrev_b16 %0, 1, %0")                                              \
    X(32, "v_add_u32 (VOP3 e64)", "v_add_u32_e64 %0, %0, %1")                                            \
    X(33, "v_bitop3 3 regs (xor3)", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")                            \
    X(34, "v_or3_b32", "v_or3_b32 %0, %0, %1, %2")                                                       \
    X(35, "v_dot4_i32_i8", "v_dot4_i32_i8 %0, %1, %2, %0")                                               \
    X(36, "v_and_b32 sdwa byte", "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1") \
    X(37, "v_add_u32 dpp row_shr", "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")                \
    X(38, "v_max_i32 dpp wave_shr", "v_max_i32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")              \
    X(39, "v_ashrrev_i32", "v_ashrrev_i32 %0, 1, %0")                                                    \
    X(40, "v_min_i32", "v_min_i32 %0, %0, %1")                                                           \
    X(41, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0")                                                       \
    X(42, "v_sub_f32", "v_sub_f32 %0, %0, %1")                                                           \
    X(43, "v_mul_f32", "v_mul_f32 %0, %0, %1")                                                           \
    X(44, "v_min_f32", "v_min_f32 %0, %0, %1")                                                           \
    X(45, "v_pk_mul_f32", "v_pk_mul_f32 %3, %3, %4")                                                     \
    X(46, "v_cndmask_b32 (sgpr mask)", "v_cndmask_b32_e64 %0, %0, %1, %5")                              \
    X(47, "v_bfe_i32", "v_bfe_i32 %0, %0, 8, 8")                                                         \
    X(48, "v_lshrrev_b32", "v_lshrrev_b32 %0, 1, %0")                                                    \
    X(49, "v_subrev_u32", "v_subrev_u32 %0, %1, %0")

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t r[8];
    uint64_t w[8];
    for (int i = 0; i < 8; ++i) { r[i] = seed * (threadIdx.x + 1) + i; w[i] = ((uint64_t)r[i] << 32) | (r[i] ^ 0x55u); }
    uint32_t c = seed | 1, d = seed * 7 + 3;
    uint64_t cw = ((uint64_t)c << 32) | d;
    unsigned long long smask = 0x5555555555555555ull ^ seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#define X(ID, NAME, TEXT) if constexpr (OP == ID) asm volatile(TEXT : "+v"(r[i]) : "v"(c), "v"(d), "v"(w[i]), "v"(cw), "s"(smask));
                OPS(X)
#undef X
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += r[i] + (uint32_t)w[i] + (uint32_t)(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, uint32_t *out) {
    const int iters = 2000, blocks = 256 * 4;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 3u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double insts_per_simd = (double)iters * 64 * 4;   // 4 waves per SIMD
    printf("%-28s %7.3f ms  %5.2f SIMD-cycles per wave instruction (2.4 GHz nominal)\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 1024 * 256 * 4);
    for (int warm = 0; warm < 200; ++warm) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, out, 2000, 3u);   // clocks
    hipDeviceSynchronize();
#define X(ID, NAME, TEXT) run<ID>(NAME, out);
    OPS(X)
#undef X
    return 0;
}
