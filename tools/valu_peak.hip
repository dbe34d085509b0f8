// valu_peak.hip -- measures the issue rate of the integer VALU instructions the DP kernels are made of.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

enum Op { ADD, BITOP3, MIN3, ALIGNBIT, ANDOR, DPP_WAVE_SHR, DPP_ROW_SHR, FMA32, PKADD16, CMP_ADDC, LSHL_OR, BFE, XOR_ };

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t r[8];
    for (int i = 0; i < 8; ++i) r[i] = seed * (threadIdx.x + 1) + i;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)r[i];
    uint32_t c = seed | 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (OP == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == XOR_) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8" : "+v"(r[i]) : "v"(c), "v"(r[(i + 1) & 7]));
                if constexpr (OP == MIN3) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(r[(i + 1) & 7]));
                if constexpr (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == ANDOR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(r[(i + 1) & 7]));
                if constexpr (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(r[i]));
                if constexpr (OP == DPP_WAVE_SHR) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
                if constexpr (OP == DPP_ROW_SHR) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
                if constexpr (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if constexpr (OP == PKADD16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == CMP_ADDC) asm volatile("v_cmp_ne_u32 vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(r[i]) : "v"(c) : "vcc");
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += r[i] + (uint32_t)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
double run(const char *name, int insts_per_iter_mult, uint32_t *out, int blocks) {
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 3u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double wave_insts = (double)blocks * 4 * iters * 64 * insts_per_iter_mult;
    double lane_ops = wave_insts * 64;
    double cyc_per_inst = (ms * 1e-3 * 2.4e9) * (256.0 * 4) / wave_insts;  // SIMD-cycles at 2.4 GHz per wave instruction
    printf("%-14s %8.3f ms  %7.2f T lane-ops/s  %5.2f cycles/wave-inst/SIMD (at 2.4 GHz nominal)\n", name, ms,
           lane_ops / (ms * 1e-3) / 1e12, cyc_per_inst);
    return ms;
}

int main() {
    uint32_t *out;
    int blocks = 256 * 8;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    run<ADD>("v_add_u32", 1, out, blocks);
    run<XOR_>("v_xor_b32", 1, out, blocks);
    run<BITOP3>("v_bitop3_b32", 1, out, blocks);
    run<MIN3>("v_min3_u32", 1, out, blocks);
    run<ALIGNBIT>("v_alignbit", 1, out, blocks);
    run<ANDOR>("v_and_or_b32", 1, out, blocks);
    run<LSHL_OR>("v_lshl_or_b32", 1, out, blocks);
    run<BFE>("v_bfe_u32", 1, out, blocks);
    run<DPP_WAVE_SHR>("dpp wave_shr", 1, out, blocks);
    run<DPP_ROW_SHR>("dpp row_shr", 1, out, blocks);
    run<FMA32>("v_fma_f32", 1, out, blocks);
    run<PKADD16>("v_pk_add_u16", 1, out, blocks);
    run<CMP_ADDC>("cmp+addc", 2, out, blocks);
    return 0;
}
