// f64 GEMM main loop with both operands k-major ([k][x]: op(A) = A^T stored K x M, B stored K x N; the ta = 1, tb = 0 form):
//   mode 0: register staging (16-byte global loads, ds_write_b128), three panels in flight -- the library's pipeline
//   mode 1: global_load_lds_dwordx4 straight into the LDS double buffer (no VGPR round trip, no ds_write)
// M = N = K = 768 (exact tiles), batch 64, and 2048 x 8.  Tuning aid.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int GK = 16;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int WT, int MODE>
__global__ __launch_bounds__(256, 2) void k(int M, int N, int K, const double* __restrict__ Ag, const double* __restrict__ Bg, double* __restrict__ Cg) {
    constexpr int HT = 32 * WT, HLD = MODE == 1 ? HT : HT + 2;
    extern __shared__ __align__(16) double lds[];
    double* As = lds;
    double* Bs = lds + 2 * GK * HLD;
    const int tm = M / HT, tn = N / HT;
    const int total = tm * tn * gridDim.y, per = (total + 7) / 8;
    const int bid = blockIdx.x + gridDim.x * blockIdx.y;
    const int lid = (bid % 8) * per + bid / 8;
    if (lid >= total) return;
    const int l = lid / (tm * tn), tt = lid % (tm * tn), i0 = (tt / tn) * HT, j0 = (tt % tn) * HT;
    const double* A = Ag + (size_t)l * M * K;     // [K][M]
    const double* B = Bg + (size_t)l * N * K;     // [K][N]
    double* C = Cg + (size_t)l * M * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wi = (wave >> 1) * 16 * WT, wj = (wave & 1) * 16 * WT;
    d4 acc[WT][WT];
    for (int a = 0; a < WT; ++a)
        for (int b = 0; b < WT; ++b) acc[a][b] = d4{0, 0, 0, 0};
    constexpr int NP = HT * GK / 512;
    double2 ra[NP], rb[NP];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            const int e = tid + 256 * h, kx = e / (HT / 2), xx = 2 * (e % (HT / 2));
            ra[h] = *reinterpret_cast<const double2*>(&A[(size_t)(k0 + kx) * M + i0 + xx]);
            rb[h] = *reinterpret_cast<const double2*>(&B[(size_t)(k0 + kx) * N + j0 + xx]);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            const int e = tid + 256 * h, kx = e / (HT / 2), xx = 2 * (e % (HT / 2));
            *reinterpret_cast<double2*>(&As[buf * GK * HLD + kx * HLD + xx]) = ra[h];
            *reinterpret_cast<double2*>(&Bs[buf * GK * HLD + kx * HLD + xx]) = rb[h];
        }
    };
    // direct loads: one wave instruction = 64 lanes x 16 bytes = 1 KB = 128 doubles = 128 / HT consecutive k-rows
    auto dma = [&](int k0, int buf) {
        constexpr int RPI = 128 / HT;                  // k-rows per instruction (1 for HT = 128, 2 for HT = 64)
        constexpr int NI = GK / RPI / 4;               // instructions per wave per operand
#pragma unroll
        for (int h = 0; h < NI; ++h) {
            const int kr = (wave + 4 * h) * RPI;       // first k-row of this instruction
            const int kx = kr + (2 * lane) / HT, xx = (2 * lane) % HT;
            __builtin_amdgcn_global_load_lds(GLB_PTR(&A[(size_t)(k0 + kx) * M + i0 + xx]), LDS_PTR(&As[buf * GK * HLD + kr * HLD]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(&B[(size_t)(k0 + kx) * N + j0 + xx]), LDS_PTR(&Bs[buf * GK * HLD + kr * HLD]), 16, 0, 0);
        }
    };
    auto mfma_panel = [&](int cur, auto&& mid) {
        const double* Ab = As + cur * GK * HLD + wi + r;
        const double* Bb = Bs + cur * GK * HLD + wj + r;
        double av[2][WT], bv[2][WT];
#pragma unroll
        for (int a = 0; a < WT; ++a) av[0][a] = Ab[q * HLD + 16 * a];
#pragma unroll
        for (int b = 0; b < WT; ++b) bv[0][b] = Bb[q * HLD + 16 * b];
#pragma unroll
        for (int st = 0; st < GK / 4; ++st) {
            if (st + 1 < GK / 4) {
#pragma unroll
                for (int a = 0; a < WT; ++a) av[(st + 1) & 1][a] = Ab[(4 * (st + 1) + q) * HLD + 16 * a];
#pragma unroll
                for (int b = 0; b < WT; ++b) bv[(st + 1) & 1][b] = Bb[(4 * (st + 1) + q) * HLD + 16 * b];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < WT; ++a)
#pragma unroll
                for (int b = 0; b < WT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st & 1][a], bv[st & 1][b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (st == 0) { mid(); __builtin_amdgcn_sched_barrier(0); }
        }
    };
    int cur = 0;
    if (MODE == 0) {
        fetch(0); stage(0);
        if (GK < K) fetch(GK);
        __syncthreads();
        for (int k0 = 0; k0 < K; k0 += GK) {
            mfma_panel(cur, [&] { if (k0 + GK < K) stage(cur ^ 1); if (k0 + 2 * GK < K) fetch(k0 + 2 * GK); });
            __syncthreads();
            cur ^= 1;
        }
    } else {
        dma(0, 0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        for (int k0 = 0; k0 < K; k0 += GK) {
            // buffer cur ^ 1 was last read in the previous panel (barrier at its end): refill it now, under this panel's MFMAs
            mfma_panel(cur, [&] { if (k0 + GK < K) dma(k0 + GK, cur ^ 1); });
            __builtin_amdgcn_s_waitcnt(0);       // vmcnt(0): this wave's direct loads have landed in LDS
            __syncthreads();
            cur ^= 1;
        }
    }
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < WT; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) C[(size_t)(i0 + wi + a * 16 + q + 4 * e) * N + j0 + wj + b * 16 + r] = acc[a][b][e];
}

template <int WT, int MODE> double run(int m, int batch, const double* A, const double* B, double* C) {
    constexpr int HT = 32 * WT, HLD = MODE == 1 ? HT : HT + 2;
    size_t lds = (size_t)4 * GK * HLD * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<WT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int tiles = (m / HT) * (m / HT);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<WT, MODE>), dim3(tiles, batch), dim3(256), lds, 0, m, m, m, A, B, C);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    double cs = 0;
    {
        double h[4];
        (void)hipMemcpy(h, C + 12345, 32, hipMemcpyDeviceToHost);
        cs = h[0] + h[1] + h[2] + h[3];
    }
    printf("m=%d x%d WT=%d mode=%d: %8.1f us/launch  %6.1f TFLOP/s  (check %.6e)\n", m, batch, WT, MODE, ms * 200.0,
           5.0 * 2.0 * m * m * m * batch / (ms * 1e-3) * 1e-12, cs);
    return cs;
}

__global__ void fill(double* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1e-3 * (double)((i * 2654435761u) % 1000) - 0.5;
}
int main() {
    const int m = 768, batch = 64;
    double *A, *B, *C;
    const size_t n = (size_t)2048 * 2048 * 8 > (size_t)m * m * batch ? (size_t)2048 * 2048 * 8 : (size_t)m * m * batch;
    (void)hipMalloc(&A, n * 8); (void)hipMalloc(&B, n * 8); (void)hipMalloc(&C, n * 8);
    hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, A, n); hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, B, n);
    run<2, 0>(m, batch, A, B, C); run<2, 1>(m, batch, A, B, C);
    run<4, 0>(m, batch, A, B, C); run<4, 1>(m, batch, A, B, C);
    run<4, 0>(2048, 8, A, B, C); run<4, 1>(2048, 8, A, B, C);
    return 0;
}
