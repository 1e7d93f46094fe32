// Ceiling experiments for the f64 MFMA GEMM inner loop on gfx950 (tuning aid, not part of the library).
//   E1  MFMA only (operands in registers)                       -> matrix-pipe ceiling at this occupancy / clock
//   E2  + LDS fragment reads (no barrier, no global)
//   E3  + one barrier per 16-deep panel
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_ceiling mfma_ceiling.hip ; run: ./mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int WT, int MODE>
__global__ __launch_bounds__(256, 2) void k(int iters, double* out, const double* in) {
    constexpr int HT = 32 * WT, HLD = HT + 2, GK = 16;
    extern __shared__ double lds[];
    double* As = lds;
    double* Bs = lds + 2 * GK * HLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wi = (wave >> 1) * 16 * WT, wj = (wave & 1) * 16 * WT;
    for (int e = tid; e < 4 * GK * HLD; e += 256) lds[e] = in[e % 1024];
    __syncthreads();
    d4 acc[WT][WT];
    for (int a = 0; a < WT; ++a)
        for (int b = 0; b < WT; ++b) acc[a][b] = d4{0, 0, 0, 0};
    double av[WT], bv[WT];
    for (int a = 0; a < WT; ++a) { av[a] = in[tid + a]; bv[a] = in[tid + 7 + a]; }
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        const double* Ab = As + cur * GK * HLD + wi + r;
        const double* Bb = Bs + cur * GK * HLD + wj + r;
#pragma unroll
        for (int kk = 0; kk < GK; kk += 4) {
            if (MODE >= 2) {
#pragma unroll
                for (int a = 0; a < WT; ++a) av[a] = Ab[(kk + q) * HLD + 16 * a];
#pragma unroll
                for (int b = 0; b < WT; ++b) bv[b] = Bb[(kk + q) * HLD + 16 * b];
            }
#pragma unroll
            for (int a = 0; a < WT; ++a)
#pragma unroll
                for (int b = 0; b < WT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (MODE >= 3) { __syncthreads(); }
        cur ^= 1;
    }
    double s = 0;
    for (int a = 0; a < WT; ++a)
        for (int b = 0; b < WT; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    out[blockIdx.x * 256 + tid] = s;
}

template <int WT, int MODE> void run(int wgs, int iters, double* out, const double* in) {
    constexpr int HT = 32 * WT;
    size_t lds = (size_t)4 * 16 * (HT + 2) * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<WT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<WT, MODE>), dim3(wgs), dim3(256), lds, 0, iters, out, in);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 10.0 * wgs * 4.0 * iters * 4 * WT * WT * 2048.0;
    printf("WT=%d mode=%d wgs=%5d iters=%d: %8.1f us/launch  %6.1f TFLOP/s\n", WT, MODE, wgs, iters, ms * 100.0, flops / (ms * 1e-3) * 1e-12);
}

int main() {
    double *out, *in;
    hipMalloc(&out, 8 << 20); hipMalloc(&in, 1 << 20);
    std::vector<double> h(131072, 1e-3);
    hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice);
    for (int wgs : {256, 512, 1024, 4096}) {
        run<2, 1>(wgs, 400, out, in); run<2, 2>(wgs, 400, out, in); run<2, 3>(wgs, 400, out, in);
        run<4, 1>(wgs, 100, out, in); run<4, 2>(wgs, 100, out, in); run<4, 3>(wgs, 100, out, in);
    }
    return 0;
}
