// Tuning aid: in-kernel time (s_memrealtime, 100 MHz) of the two 64 x 64 diagonal-block factorisations of cholesky.hip
// (chol64_lds: four waves, register tiles; chol64_wave: one wave, a row per lane) and of the triangular inverse, on `nwg`
// workgroups.    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I svgp-vae_amd/csrc tools/micro/chol_probe.hip -L svgp-vae_amd -lsvgpvae_hip -o chol_probe
#include "cholesky.hip"
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_probe(int mode, unsigned long long* out, double* sink) {
    extern __shared__ __align__(16) unsigned char diag_lds[];
    real (*colk)[CB] = reinterpret_cast<real (*)[CB]>(diag_lds);
    real* rdiag = reinterpret_cast<real*>(diag_lds + 2 * CB * sizeof(real));
    real (*Ls)[CLD] = reinterpret_cast<real (*)[CLD]>(diag_lds + 3 * CB * sizeof(real));
    real (*Xs)[CLD] = reinterpret_cast<real (*)[CLD]>(diag_lds + (3 * CB + CB * CLD) * sizeof(real));
    for (int e = threadIdx.x; e < CB * CB; e += blockDim.x) {
        const int i = e / CB, j = e % CB;
        Ls[i][j] = (i == j ? 70.0 : 0.0) + 1.0 / (1.0 + (i > j ? i - j : j - i));
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    chol64(Ls, colk, rdiag, mode);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    trinv64_lds(Ls, Xs, rdiag);
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = t2 - t1; }
    sink[blockIdx.x * 256 + threadIdx.x] = Ls[threadIdx.x & 63][threadIdx.x >> 2] + Xs[threadIdx.x & 63][threadIdx.x >> 2];
}
int main() {
    const int nwg = 65;
    unsigned long long* out; double* sink;
    hipMalloc(&out, nwg * 16); hipMalloc(&sink, nwg * 256 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DIAG_LDS_BYTES);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(256), DIAG_LDS_BYTES, 0, mode, out, sink);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(2 * nwg);
            hipMemcpy(h.data(), out, nwg * 16, hipMemcpyDeviceToHost);
            std::vector<double> hs(256);
            hipMemcpy(hs.data(), sink, 256 * 8, hipMemcpyDeviceToHost);
            printf("mode %d rep %d: chol %.2f us  trinv %.2f us   (check %.12g)\n", mode, rep, h[0] / 100.0, h[1] / 100.0, hs[77]);
        }
    return 0;
}
