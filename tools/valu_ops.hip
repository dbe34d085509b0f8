// valu_ops.hip -- SIMD cycles per wave instruction for the instruction forms hipcc picks in the bit-parallel kernels
// (8 independent chains per wave, 4 waves per SIMD: issue-bound, not latency-bound).
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -w tools/valu_ops.hip -o /tmp/valu_ops && /tmp/valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t r[8];
    uint64_t w[8];
    for (int i = 0; i < 8; ++i) { r[i] = seed * (threadIdx.x + 1) + i; w[i] = ((uint64_t)r[i] << 32) | (r[i] ^ 0x55u); }
    uint32_t c = seed | 1, d = seed * 7 + 3;
    unsigned long long smask = 0x5555555555555555ull ^ seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8" : "+v"(r[i]) : "v"(c), "v"(d));
                if constexpr (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
                if constexpr (OP == 4) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(w[i]));
                if constexpr (OP == 5) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(d));
                if constexpr (OP == 6) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r[i]) : "v"(c) : "vcc");
                if constexpr (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 8) asm volatile("v_bfe_u32 %0, %0, 4, 4" : "+v"(r[i]));
                if constexpr (OP == 9) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(d));
                if constexpr (OP == 10) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(d));
                if constexpr (OP == 11) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0xe8" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 12) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 13) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 14) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 15) asm volatile("v_or_b32 %0, %1, %0" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 16) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(c) : "vcc");
                if constexpr (OP == 17) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r[i]) : "v"(c) : "vcc");
                if constexpr (OP == 18) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[i]));
                if constexpr (OP == 19) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 20) asm volatile("v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 21) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(d));
                if constexpr (OP == 22) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(d));
                if constexpr (OP == 23) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 24) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 25) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 26) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 29) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(c) : "vcc");
                if constexpr (OP == 30) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "s"(smask));
                if constexpr (OP == 31) asm volatile("v_cmp_lt_u32_e64 %2, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "s"(smask));
                if constexpr (OP == 32) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r[i]), "v"(c) : "vcc");
                if constexpr (OP == 33) asm volatile("v_cmp_lt_i64 vcc, %0, %1" : : "v"(w[i]), "v"(w[(i + 1) & 7]) : "vcc");
                if constexpr (OP == 34) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 27) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r[i]) : "v"(r[(i + 3) & 7]));
                if constexpr (OP == 28) asm volatile("v_and_b32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r[i]) : "v"(r[(i + 3) & 7]), "v"(c));
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += r[i] + (uint32_t)w[i] + (uint32_t)(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, int per, uint32_t *out) {
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
    double insts_per_simd = (double)iters * 64 * per * 4;   // 4 waves per SIMD
    printf("%-22s %7.3f ms  %5.2f SIMD-cycles per wave instruction (2.4 GHz nominal)\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 1024 * 256 * 4);
    run<0>("v_xor_b32", 1, out); run<14>("v_and_b32", 1, out); run<15>("v_or_b32", 1, out); run<18>("v_lshlrev_b32", 1, out);
    run<1>("v_bitop3 (3 regs)", 1, out); run<11>("v_bitop3 (2 regs)", 1, out);
    run<2>("v_alignbit_b32", 1, out); run<12>("v_alignbyte_b32", 1, out); run<13>("v_lshl_or_b32", 1, out);
    run<3>("v_lshl_add_u64", 1, out); run<4>("v_lshrrev_b64", 1, out); run<5>("v_mad_u32_u24", 1, out);
    run<6>("v_add_co+v_addc_co", 2, out); run<16>("v_add_co_u32", 1, out); run<17>("v_addc_co_u32", 1, out);
    run<19>("v_add_u32_sdwa byte", 1, out); run<20>("v_add_u32_sdwa sext", 1, out); run<21>("v_max3_i32", 1, out);
    run<22>("v_perm_b32", 1, out); run<23>("v_max_i32", 1, out); run<24>("v_add_u32", 1, out);
    run<25>("v_pk_max_i16", 1, out); run<26>("v_pk_add_i16", 1, out);
    run<27>("v_mov_b32_dpp wave_shr", 1, out); run<28>("v_and_b32_dpp wave_shr", 1, out);
    run<7>("v_cndmask_b32 vcc (vcc never written)", 1, out); run<29>("v_cmp + v_cndmask vcc", 2, out);
    run<30>("v_cndmask_b32_e64 sgpr", 1, out); run<31>("v_cmp_e64 + v_cndmask_e64", 2, out);
    run<32>("v_cmp_lt_u32 vcc", 1, out); run<33>("v_cmp_lt_i64 vcc", 1, out); run<34>("v_min_u32", 1, out); run<8>("v_bfe_u32", 1, out); run<9>("v_and_or_b32", 1, out); run<10>("v_add3_u32", 1, out);
    return 0;
}
