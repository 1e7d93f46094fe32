// Where the f64 GEMM loses time: the library's 64 / 128-tile loop (A [i][k], B [j][k], i.e. ta = 0, tb = 1) with pieces
// switched off.  MODE 3: MFMA + LDS reads + barrier; 4: + LDS staging stores (registers, no global loads);
// 5: + global loads (the full main loop, no epilogue); 6: + C stores (the full kernel, exact multiples of the tile).
// Problem: M = N = K = 768 (exact tiles), batch 64.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int GK = 16;

template <int WT, int MODE, bool VEC>
__global__ __launch_bounds__(256, 2) void k(int M, int N, int K, const double* __restrict__ Ag, const double* __restrict__ Bg, double* __restrict__ Cg) {
    constexpr int HT = 32 * WT, HLD = HT + 2, NH = HT / 16;
    extern __shared__ double lds[];
    double* As = lds;
    double* Bs = lds + 2 * GK * HLD;
    const int tm = M / HT, tn = N / HT;
    const int total = tm * tn * gridDim.y, per = (total + 7) / 8;
    const int bid = blockIdx.x + gridDim.x * blockIdx.y;
    const int lid = (bid % 8) * per + bid / 8;
    if (lid >= total) return;
    const int l = lid / (tm * tn), tt = lid % (tm * tn), i0 = (tt / tn) * HT, j0 = (tt % tn) * HT;
    const double* A = Ag + (size_t)l * M * K;
    const double* B = Bg + (size_t)l * N * K;
    double* C = Cg + (size_t)l * M * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wi = (wave >> 1) * 16 * WT, wj = (wave & 1) * 16 * WT;
    d4 acc[WT][WT];
    for (int a = 0; a < WT; ++a)
        for (int b = 0; b < WT; ++b) acc[a][b] = d4{0, 0, 0, 0};
    double ra[NH], rb[NH];
    for (int h = 0; h < NH; ++h) { ra[h] = 1e-3 * tid; rb[h] = 1e-3 * h; }
    auto fetch = [&](int k0) {
        if (MODE < 5) return;
        if (VEC) {
            // 2 consecutive k per thread (16-byte loads): k = 2 (tid & 7), x = (tid >> 3) + 32 h
#pragma unroll
            for (int h = 0; h < NH / 2; ++h) {
                const int kx = 2 * (tid & 7), x = (tid >> 3) + 32 * h;
                const double2 va = *reinterpret_cast<const double2*>(&A[(size_t)(i0 + x) * K + k0 + kx]);
                const double2 vb = *reinterpret_cast<const double2*>(&B[(size_t)(j0 + x) * K + k0 + kx]);
                ra[2 * h] = va.x; ra[2 * h + 1] = va.y; rb[2 * h] = vb.x; rb[2 * h + 1] = vb.y;
            }
        } else {
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int kx = tid & 15, x = (tid >> 4) + 16 * h;
                ra[h] = A[(size_t)(i0 + x) * K + k0 + kx];
                rb[h] = B[(size_t)(j0 + x) * K + k0 + kx];
            }
        }
    };
    auto stage = [&](int buf) {
        if (MODE < 4) return;
        double* Ad = As + buf * GK * HLD;
        double* Bd = Bs + buf * GK * HLD;
        if (VEC) {
#pragma unroll
            for (int h = 0; h < NH / 2; ++h) {
                const int kx = 2 * (tid & 7), x = (tid >> 3) + 32 * h;
                Ad[kx * HLD + x] = ra[2 * h]; Ad[(kx + 1) * HLD + x] = ra[2 * h + 1];
                Bd[kx * HLD + x] = rb[2 * h]; Bd[(kx + 1) * HLD + x] = rb[2 * h + 1];
            }
        } else {
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                Ad[(tid & 15) * HLD + (tid >> 4) + 16 * h] = ra[h];
                Bd[(tid & 15) * HLD + (tid >> 4) + 16 * h] = rb[h];
            }
        }
    };
    for (int e = tid; e < 4 * GK * HLD; e += 256) lds[e] = 1e-3;
    fetch(0); stage(0);
    if (MODE >= 7) {
        if (GK < K) fetch(GK);
        __syncthreads();
        int cur = 0;
        for (int k0 = 0; k0 < K; k0 += GK) {
            const double* Ab = As + cur * GK * HLD + wi + r;
            const double* Bb = Bs + cur * GK * HLD + wj + r;
            double av[2][WT], bv[2][WT];
            if (MODE == 8 && k0 + GK < K) stage(cur ^ 1);
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
                if (MODE == 7 && st == 0) {
                    if (k0 + GK < K) stage(cur ^ 1);
                    if (k0 + 2 * GK < K) fetch(k0 + 2 * GK);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MODE == 8 && k0 + 2 * GK < K) fetch(k0 + 2 * GK);
            __syncthreads();
            cur ^= 1;
        }
    } else {
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < K; k0 += GK) {
        const bool more = k0 + GK < K;
        if (more) fetch(k0 + GK);
        const double* Ab = As + cur * GK * HLD + wi + r;
        const double* Bb = Bs + cur * GK * HLD + wj + r;
#pragma unroll
        for (int kk = 0; kk < GK; kk += 4) {
            double av[WT], bv[WT];
#pragma unroll
            for (int a = 0; a < WT; ++a) av[a] = Ab[(kk + q) * HLD + 16 * a];
#pragma unroll
            for (int b = 0; b < WT; ++b) bv[b] = Bb[(kk + q) * HLD + 16 * b];
#pragma unroll
            for (int a = 0; a < WT; ++a)
#pragma unroll
                for (int b = 0; b < WT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (more) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    }
    if (MODE >= 6) {
#pragma unroll
        for (int a = 0; a < WT; ++a)
#pragma unroll
            for (int b = 0; b < WT; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) C[(size_t)(i0 + wi + a * 16 + q + 4 * e) * N + j0 + wj + b * 16 + r] = acc[a][b][e];
    } else {
        double s = 0;
        for (int a = 0; a < WT; ++a)
            for (int b = 0; b < WT; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
        if (s == 12345.0) C[tid] = s;
    }
}

template <int WT, int MODE, bool VEC> void run(int m, int batch, const double* A, const double* B, double* C) {
    constexpr int HT = 32 * WT;
    size_t lds = (size_t)4 * GK * (HT + 2) * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<WT, MODE, VEC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int tiles = (m / HT) * (m / HT);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<WT, MODE, VEC>), dim3(tiles, batch), dim3(256), lds, 0, m, m, m, A, B, C);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("m=%d x%d WT=%d mode=%d vec=%d: %8.1f us/launch  %6.1f TFLOP/s\n", m, batch, WT, MODE, (int)VEC, ms * 200.0,
           5.0 * 2.0 * m * m * m * batch / (ms * 1e-3) * 1e-12);
}

__global__ void fill(double* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1e-3 * (double)((i * 2654435761u) % 1000) - 0.5;
}
int main() {
    const int m = 768, batch = 64;
    double *A, *B, *C;
    (void)hipMalloc(&A, (size_t)m * m * batch * 8); (void)hipMalloc(&B, (size_t)m * m * batch * 8); (void)hipMalloc(&C, (size_t)m * m * batch * 8);
    hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, A, (size_t)m * m * batch); hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, B, (size_t)m * m * batch);
    run<2, 3, false>(m, batch, A, B, C); run<2, 6, false>(m, batch, A, B, C); run<2, 6, true>(m, batch, A, B, C);
    run<2, 7, false>(m, batch, A, B, C); run<2, 7, true>(m, batch, A, B, C); run<2, 8, true>(m, batch, A, B, C);
    run<4, 3, false>(m, batch, A, B, C); run<4, 6, false>(m, batch, A, B, C); run<4, 6, true>(m, batch, A, B, C);
    run<4, 7, false>(m, batch, A, B, C); run<4, 7, true>(m, batch, A, B, C); run<4, 8, true>(m, batch, A, B, C);
    run<4, 6, true>(2048, 8, A, B, C); run<4, 7, true>(2048, 8, A, B, C); run<4, 8, true>(2048, 8, A, B, C);
    return 0;
}
