// Measuring tool: what a synchronous call costs around its kernel -- launch + stream synchronize of an empty kernel, plain and as a
// one-node hipGraph, and with a host-mapped flag polled instead of the stream. usage: launch_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_empty(volatile unsigned *flag, unsigned value) { if (flag && threadIdx.x == 0 && blockIdx.x == 0) *flag = value; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *flag; CK(hipHostMalloc(&flag, 64, hipHostMallocMapped)); *flag = 0;
    unsigned *dflag; CK(hipHostGetDevicePointer((void **)&dflag, flag, 0));
    const int n = 2000;
    for (int grid : {1, 512}) {
        for (int i = 0; i < 200; ++i) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s, (volatile unsigned *)nullptr, 0u); CK(hipStreamSynchronize(s)); }
        double t0 = now();
        for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s, (volatile unsigned *)nullptr, 0u); CK(hipStreamSynchronize(s)); }
        printf("grid %3d  launch + hipStreamSynchronize      %.2f us\n", grid, (now() - t0) / n);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s, (volatile unsigned *)nullptr, 0u);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 200; ++i) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        t0 = now();
        for (int i = 0; i < n; ++i) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        printf("grid %3d  hipGraphLaunch + hipStreamSynchronize %.2f us\n", grid, (now() - t0) / n);
        t0 = now();
        for (int i = 0; i < n; ++i) {
            hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s, (volatile unsigned *)dflag, (unsigned)(i + 1));
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (unsigned)(i + 1)) { }
        }
        printf("grid %3d  launch + poll a host-mapped word     %.2f us\n", grid, (now() - t0) / n);
        CK(hipStreamSynchronize(s));
    }
    return 0;
}
