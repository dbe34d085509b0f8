// gotoh_sched.hip -- what the VALU schedule of nwprofile.hip's Gotoh cells costs by itself: the asm planes of run_pass over
// registers only (no LDS, no DPP, no loop overhead), single-wave workgroups at 1 / 2 / 3 waves per SIMD.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -w tools/gotoh_sched.hip -o /tmp/gotoh_sched && /tmp/gotoh_sched
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define SWH_NWP_SCORE_ADD(DST, BASE, WORD, BYTE) \
    "v_add_u32_sdwa " DST ", " BASE ", sext(" WORD ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" BYTE "\n"
#define SWH_NWP_CELL(MAX, LEFT, FNEXT, HNEXT, TCUR, FCUR, TNEXT, WORD, BYTE, HOUT)                                      \
    MAX " %[e], " LEFT ", %[e]\n" MAX " " FNEXT ", " HNEXT ", " FNEXT "\n"                                             \
        "v_max3_i32 %[tmp], " TCUR ", %[e], " FCUR "\n" SWH_NWP_SCORE_ADD(TNEXT, HNEXT, WORD, BYTE) "v_add_u32 " HOUT ", %[tmp], %[c]\n"
#define SWH_NWP_PLANE(MAX)                                                                                              \
    SWH_NWP_CELL(MAX, "%[left]", "%[f1]", "%[h1]", "%[t0]", "%[f0]", "%[t2]", "%[s0]", "2", "%[h0]")                     \
    SWH_NWP_CELL(MAX, "%[h0]", "%[f2]", "%[h2]", "%[t1]", "%[f1]", "%[t3]", "%[s0]", "3", "%[h1]")                       \
    SWH_NWP_CELL(MAX, "%[h1]", "%[f3]", "%[h3]", "%[t2]", "%[f2]", "%[t4]", "%[s1]", "0", "%[h2]")                       \
    SWH_NWP_CELL(MAX, "%[h2]", "%[f4]", "%[hn]", "%[t3]", "%[f3]", "%[t5]", "%[s1]", "1", "%[h3]")

// the same five operations per cell, chain back to back (what hipcc emits when left alone, minus its batching of the f)
#define CELL_SERIAL(MAX, LEFT, FNEXT, HNEXT, TCUR, FCUR, TNEXT, WORD, BYTE, HOUT)                                       \
    MAX " " FNEXT ", " HNEXT ", " FNEXT "\n" SWH_NWP_SCORE_ADD(TNEXT, HNEXT, WORD, BYTE) MAX " %[e], " LEFT ", %[e]\n"     \
        "v_max3_i32 %[tmp], " TCUR ", %[e], " FCUR "\nv_add_u32 " HOUT ", %[tmp], %[c]\n"
#define PLANE_SERIAL(MAX)                                                                                               \
    CELL_SERIAL(MAX, "%[left]", "%[f1]", "%[h1]", "%[t0]", "%[f0]", "%[t2]", "%[s0]", "2", "%[h0]")                      \
    CELL_SERIAL(MAX, "%[h0]", "%[f2]", "%[h2]", "%[t1]", "%[f1]", "%[t3]", "%[s0]", "3", "%[h1]")                        \
    CELL_SERIAL(MAX, "%[h1]", "%[f3]", "%[h3]", "%[t2]", "%[f2]", "%[t4]", "%[s1]", "0", "%[h2]")                        \
    CELL_SERIAL(MAX, "%[h2]", "%[f4]", "%[hn]", "%[t3]", "%[f3]", "%[t5]", "%[s1]", "1", "%[h3]")

#define RUN_PLANE(TEXT)                                                                                                                \
    asm volatile(TEXT : [h0] "+v"(H[k0]), [h1] "+v"(H[k0 + 1]), [h2] "+v"(H[k0 + 2]), [h3] "+v"(H[k0 + 3]), [f1] "+v"(F[k0 + 1]),       \
                 [f2] "+v"(F[k0 + 2]), [f3] "+v"(F[k0 + 3]), [f4] "+v"(F[kn]), [e] "+v"(e), [t2] "=&v"(t[k0 + 2]), [t3] "=&v"(t[k0 + 3]), \
                 [t4] "=&v"(t[kn]), [t5] "=&v"(t[kn + 1]), [tmp] "=&v"(h3)                                                             \
                 : [left] "v"(left), [f0] "v"(F[k0]), [hn] "v"(H[kn]), [t0] "v"(t[k0]), [t1] "v"(t[k0 + 1]), [s0] "v"(s[k0 / 4]),      \
                   [s1] "v"(s[kn / 4]), [c] "v"(c))

template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, uint32_t seed) {
    constexpr int W = 16;
    int H[W], F[W], t[W + 2], s[W / 4];
    for (int i = 0; i < W; ++i) { H[i] = 0x8000 + (int)((seed * (threadIdx.x + 3) + i) & 255); F[i] = H[i] - 5; }
    for (int i = 0; i < W / 4; ++i) s[i] = (int)(seed * 0x01030507u + i);
    t[0] = H[3]; t[1] = H[5];
    int e = 0x7000, left = 0x8001, c = -3, h3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int pl = 0; pl < W / 4; ++pl) {
            const int k0 = 4 * pl, kn = 4 * ((pl + 1) % (W / 4));
            if constexpr (MODE == 0) RUN_PLANE(SWH_NWP_PLANE("v_max_u16"));
            if constexpr (MODE == 1) RUN_PLANE(SWH_NWP_PLANE("v_max_i32"));
            if constexpr (MODE == 2) RUN_PLANE(PLANE_SERIAL("v_max_u16"));
            if constexpr (MODE == 3) RUN_PLANE(PLANE_SERIAL("v_max_i32"));
            left = H[k0 + 3];
        }
    }
    uint32_t sum = (uint32_t)e;
    for (int i = 0; i < W; ++i) sum += (uint32_t)H[i] + (uint32_t)F[i];
    out[blockIdx.x * 64 + threadIdx.x] = sum + (uint32_t)t[0] + (uint32_t)t[1];
}

template <int MODE>
void row(const char *name, uint32_t *out) {
    printf("%-44s", name);
    for (int waves : {1, 2, 3, 4}) {
        const int iters = 4000, blocks = 256 * 4 * waves;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 10, 3u);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 3u);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("  %6.2f", ms * 1e-3 * 2.4e9 / ((double)iters * 16 * waves));   // SIMD cycles per wave-cell
    }
    printf("\n");
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 4096 * 64 * 4);
    for (int warm = 0; warm < 100; ++warm) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(64), 0, 0, out, 4000, 3u);   // clocks
    hipDeviceSynchronize();
    printf("SIMD cycles per Gotoh cell of one wave (2.4 GHz nominal) at 1 / 2 / 3 / 4 waves per SIMD\n");
    row<0>("interleaved, v_max_u16", out);
    row<1>("interleaved, v_max_i32", out);
    row<2>("chain back to back, v_max_u16", out);
    row<3>("chain back to back, v_max_i32", out);
    return 0;
}
