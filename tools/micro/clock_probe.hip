// Shader clock seen by a small latency-bound launch vs. a chip-filling one: s_memtime (core clock) against
// s_memrealtime (constant 100 MHz) around a dependent FMA chain.  Tuning aid.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int n, double* out, unsigned long long* clk) {
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0000001;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, y, 1e-12);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
    double* out; unsigned long long* clk;
    (void)hipMalloc(&out, 8 << 20); (void)hipMalloc(&clk, 1 << 20);
    unsigned long long h[2];
    for (int wgs : {1, 16, 256, 2048}) {
        for (int n : {2000, 20000, 200000}) {
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, n, out, clk);
                (void)hipDeviceSynchronize();
            }
            (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            printf("wgs=%5d chain=%7d: core ticks %9llu, 100MHz ticks %7llu -> %.0f MHz, %.2f core cycles per dependent f64 FMA, %.1f us\n", wgs, n, h[0], h[1],
                   100.0 * h[0] / h[1], (double)h[0] / n, h[1] / 100.0);
        }
    }
    // back-to-back short launches (like the training step): 200 launches of 16 WGs x chain 2000
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, 0, 2000, out, clk);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("after 200 back-to-back short launches: %.0f MHz, %.2f cycles per FMA\n", 100.0 * h[0] / h[1], (double)h[0] / 2000);
    return 0;
}
