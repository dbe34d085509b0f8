#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    unsigned s0 = 0x07060504u, s1 = 0x03020100u;   // bytes named by their own index if sel 0-3 -> s1, 4-7 -> s0
    out[0] = __builtin_amdgcn_perm(s0, s1, 0x00010203u);
    out[1] = __builtin_amdgcn_perm(s0, s1, 0x04050607u);
    out[2] = __builtin_amdgcn_perm(s0, s1, 0x0c0c0c07u);
    out[3] = __builtin_amdgcn_perm(s0, s1, 0x0d0c0003u);
    unsigned r = 0x80FF7F01u;
    int a = (int)(signed char)(r >> 24), b = (int)(signed char)(r >> 16);
    out[4] = (unsigned)(a + 1000); out[5] = (unsigned)(b + 1000);
}
int main() { unsigned *d, h[6]; hipMalloc(&d, 24); hipLaunchKernelGGL(k, 1, 1, 0, 0, d); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  for (int i = 0; i < 6; ++i) printf("%08x\n", h[i]); return 0; }
