// valu_chain.hip -- how much of the VALU a SIMD delivers as a function of resident waves and of the instruction-level
// parallelism inside one wave (dependent chains). Explains why a kernel limited to 1-2 waves/SIMD by its LDS footprint
// cannot reach the integer peak with a serial recurrence.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/valu_chain.hip -o /tmp/valu_chain && /tmp/valu_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int ILP, int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
    extern __shared__ char smem[];
    uint32_t r[8];
    for (int i = 0; i < 8; ++i) r[i] = seed * (threadIdx.x + 1) + i;
    uint32_t c = seed | 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 64 / ILP; ++rep) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if constexpr (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0xe8" : "+v"(r[i]) : "v"(c));
                if constexpr (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(r[i]) : "v"(c));
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 0x12345) ((uint32_t *)smem)[threadIdx.x] = s;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP, int OP>
void run(uint32_t *out, int waves_per_simd) {
    const int iters = 4000;
    size_t lds = 160 * 1024 / waves_per_simd - 1024;   // one 256-thread block = 1 wave per SIMD
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k<ILP, OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int blocks = 256 * waves_per_simd;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<ILP, OP>), dim3(blocks), dim3(256), lds, 0, out, 10, 3u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<ILP, OP>), dim3(blocks), dim3(256), lds, 0, out, iters, 3u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double insts_per_wave = (double)iters * (64 / ILP) * ILP;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("op %d  waves/SIMD %d  ILP %d : %7.3f ms  %5.2f cycles per instruction of one wave, %5.2f SIMD-cycles per wave-inst\n",
           OP, waves_per_simd, ILP, ms, cyc / insts_per_wave, cyc / (insts_per_wave * waves_per_simd));
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 3, 4, 8}) {
        run<1, 0>(out, w); run<2, 0>(out, w); run<4, 0>(out, w); run<8, 0>(out, w);
    }
    for (int w : {1, 2, 4}) { run<1, 1>(out, w); run<2, 1>(out, w); run<4, 1>(out, w); }
    for (int w : {1, 2, 4}) { run<1, 2>(out, w); run<4, 2>(out, w); }
    return 0;
}
