// f64 / f32 FMA issue vs latency for ONE wave per SIMD: ILP independent dependent-chains per lane.  Also ds_bpermute and v_rcp_f64.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP, typename T>
__global__ void k(int n, T* out, unsigned long long* clk) {
    T x[ILP];
    for (int j = 0; j < ILP; ++j) x[j] = T(1.0) + T(threadIdx.x + j) * T(1e-6);
    const T y = T(1.0000001);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int j = 0; j < ILP; ++j) x[j] = __builtin_fma(x[j], y, T(1e-12));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[0] = c1 - c0;
    T s = 0;
    for (int j = 0; j < ILP; ++j) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_rcp(int n, double* out, unsigned long long* clk) {
    double x = 1.5 + threadIdx.x * 1e-6;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) x = __builtin_amdgcn_rcp(x) + 0.25;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[0] = c1 - c0;
    out[threadIdx.x] = x;
}
__global__ void k_perm(int n, double* out, unsigned long long* clk) {
    double x = 1.5 + threadIdx.x * 1e-6;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) x = __shfl(x, (threadIdx.x * 7 + i) & 63, 64);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[0] = c1 - c0;
    out[threadIdx.x] = x;
}
__global__ void k_mfma(int n, double* out, unsigned long long* clk) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc = {0, 0, 0, 0};
    double a = 1e-3 * threadIdx.x, b = 1.0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[0] = c1 - c0;
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <typename F> void run(const char* name, int n, int per, F f, unsigned long long* clk) {
    unsigned long long h;
    for (int r = 0; r < 2; ++r) { f(); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-34s %7.2f core cycles per op (%d ops per trip)\n", name, (double)h / n / per, per);
}
int main() {
    double* out; unsigned long long* clk;
    (void)hipMalloc(&out, 8 << 20); (void)hipMalloc(&clk, 64);
    const int n = 20000;
    run("f64 fma, 1 wave, ILP 1", n, 1, [&] { hipLaunchKernelGGL((k<1, double>), dim3(1), dim3(64), 0, 0, n, out, clk); }, clk);
    run("f64 fma, 1 wave, ILP 2", n, 2, [&] { hipLaunchKernelGGL((k<2, double>), dim3(1), dim3(64), 0, 0, n, out, clk); }, clk);
    run("f64 fma, 1 wave, ILP 4", n, 4, [&] { hipLaunchKernelGGL((k<4, double>), dim3(1), dim3(64), 0, 0, n, out, clk); }, clk);
    run("f64 fma, 1 wave, ILP 8", n, 8, [&] { hipLaunchKernelGGL((k<8, double>), dim3(1), dim3(64), 0, 0, n, out, clk); }, clk);
    run("f64 fma, 1 wave, ILP 16", n, 16, [&] { hipLaunchKernelGGL((k<16, double>), dim3(1), dim3(64), 0, 0, n, out, clk); }, clk);
    run("f64 fma, 4 waves (1/SIMD), ILP 1", n, 1, [&] { hipLaunchKernelGGL((k<1, double>), dim3(1), dim3(256), 0, 0, n, out, clk); }, clk);
    run("f64 fma, 8 waves (2/SIMD), ILP 1", n, 1, [&] { hipLaunchKernelGGL((k<1, double>), dim3(1), dim3(512), 0, 0, n, out, clk); }, clk);
    run("f64 fma, 16 waves (4/SIMD), ILP 1", n, 1, [&] { hipLaunchKernelGGL((k<1, double>), dim3(1), dim3(1024), 0, 0, n, out, clk); }, clk);
    float* outf = reinterpret_cast<float*>(out);
    run("f32 fma, 1 wave, ILP 1", n, 1, [&] { hipLaunchKernelGGL((k<1, float>), dim3(1), dim3(64), 0, 0, n, outf, clk); }, clk);
    run("f32 fma, 1 wave, ILP 4", n, 4, [&] { hipLaunchKernelGGL((k<4, float>), dim3(1), dim3(64), 0, 0, n, outf, clk); }, clk);
    run("f32 fma, 1 wave, ILP 8", n, 8, [&] { hipLaunchKernelGGL((k<8, float>), dim3(1), dim3(64), 0, 0, n, outf, clk); }, clk);
    run("v_rcp_f64 + add chain", n, 1, [&] { hipLaunchKernelGGL(k_rcp, dim3(1), dim3(64), 0, 0, n, out, clk); }, clk);
    run("ds_bpermute f64 chain", n, 1, [&] { hipLaunchKernelGGL(k_perm, dim3(1), dim3(64), 0, 0, n, out, clk); }, clk);
    run("dependent mfma f64 16x16x4", n, 1, [&] { hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, n, out, clk); }, clk);
    return 0;
}
