// The single-wave 32 x 32 Gauss-Jordan sweep: the 8 x 8-lane ds_bpermute form (rounds 1-4, `old` below) against the
// 4 x 16-lane DPP form of svgp-vae_amd/csrc/sweep32.hpp.  Checks that both give the SAME BITS (inverse and pivots) on a
// well-conditioned and on a kernel-like (K + 1e-6 I) block, and times `REPS` back-to-back sweeps with s_memrealtime (100 MHz)
// on one workgroup alone and on 256 / 1024 workgroups (one wave of each works, as in the library's callers).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I svgp-vae_amd/csrc tools/micro/sweep32_probe.hip -o tools/micro/sweep32_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "sweep32.hpp"

typedef double real;
#define NB 32
#define REPS 64

namespace old {
__device__ __forceinline__ real fast_rcp_la(real x) {
    real r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, real(1)), r, r);
    r = fma(fma(-x, r, real(1)), r, r);
    return r;
}
__device__ __forceinline__ real readlane_f64_la(real v, int src) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, src), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
// the sweep of linalg.hip / gp_kernels.hip up to round 4, verbatim arithmetic
__device__ __forceinline__ real sweep(const real (*P)[NB + 1], real* __restrict__ Pinv) {
    const int lane = threadIdx.x, bi = lane >> 3, bj = lane & 7;
    real a[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) a[r][c] = P[bi * 4 + r][bj * 4 + c];
    real mypiv = 1;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int kq = k / 4, kr = k % 4;
        real rowk[4], colk[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) rowk[c] = __shfl(a[kr][c], kq * 8 + bj, 64);
#pragma unroll
        for (int r = 0; r < 4; ++r) colk[r] = __shfl(a[r][kr], bi * 8 + kq, 64);
        const real piv = readlane_f64_la(a[kr][kr], kq * 9);
        const real ipiv = fast_rcp_la(piv);
        if (lane == k) mypiv = piv;
        real rkj[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) rkj[c] = (bj * 4 + c == k) ? ipiv : rowk[c] * ipiv;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool ik = (bi * 4 + r == k), jk = (bj * 4 + c == k);
                a[r][c] = ik ? rkj[c] : ((jk ? real(0) : a[r][c]) - colk[r] * rkj[c]);
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) Pinv[(bi * 4 + r) * NB + bj * 4 + c] = a[r][c];
    return mypiv;
}
}  // namespace old

// out: (2 modes... ) inverse (32 x 32) + pivots (64) per workgroup 0; ticks per workgroup
__global__ __launch_bounds__(256) void k_probe(int mode, const real* __restrict__ A, real* __restrict__ inv, real* __restrict__ piv,
                                               unsigned long long* __restrict__ ticks) {
    __shared__ real P[NB][NB + 1];
    __shared__ real Q[NB * NB];
    for (int t = threadIdx.x; t < NB * NB; t += blockDim.x) P[t / NB][t % NB] = A[t];
    __syncthreads();
    if (threadIdx.x < 64) {
        real mp = 0;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int rep = 0; rep < REPS; ++rep) {
            if (mode == 0) {
                mp = old::sweep(P, Q);
            } else {
                mp = sweep32::gauss_jordan_32(threadIdx.x, NB, [&](int i, int j) { return P[i][j]; },
                                              [&](int i, int j, real v) { Q[i * NB + j] = v; });
            }
            __builtin_amdgcn_s_waitcnt(0);      // the stores of this sweep before the next one's loads (same addresses)
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
        if (blockIdx.x == 0) piv[threadIdx.x] = mp;
    }
    __syncthreads();
    if (blockIdx.x == 0)
        for (int t = threadIdx.x; t < NB * NB; t += blockDim.x) inv[t] = Q[t];
}

static void make_block(int kind, std::vector<double>& A) {
    A.assign(NB * NB, 0.0);
    if (kind == 0) {                                   // X X^T / 40 + 0.05 I
        std::vector<double> X(NB * 40);
        unsigned s = 12345u;
        for (auto& x : X) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffff) / 32768.0 - 1.0; }
        for (int i = 0; i < NB; ++i)
            for (int j = 0; j < NB; ++j) {
                double acc = 0;
                for (int k = 0; k < 40; ++k) acc += X[i * 40 + k] * X[j * 40 + k];
                A[i * NB + j] = acc / 40 + (i == j ? 0.05 : 0.0);
            }
    } else {                                           // periodic kernel on 16 angles x 2 objects + jitter 1e-6 (config-2-like)
        for (int i = 0; i < NB; ++i)
            for (int j = 0; j < NB; ++j) {
                const double ti = 2 * M_PI * (i / 2) / 16.0, tj = 2 * M_PI * (j / 2) / 16.0;
                const double oi = (i % 2) ? 0.9 : 1.1, oj = (j % 2) ? 0.9 : 1.1, d = std::sin(0.5 * (ti - tj));
                A[i * NB + j] = std::exp(-2.0 * d * d) * (oi * oj + 0.3 * ((i % 2) == (j % 2) ? 1.0 : -0.2)) + (i == j ? 1e-6 : 0.0);
            }
    }
}

int main() {
    real *dA, *dinv, *dpiv;
    unsigned long long* dt;
    const int maxwg = 1024;
    hipMalloc(&dA, NB * NB * 8); hipMalloc(&dinv, NB * NB * 8); hipMalloc(&dpiv, 64 * 8); hipMalloc(&dt, maxwg * 8);
    int bad = 0;
    for (int kind = 0; kind < 2; ++kind) {
        std::vector<double> A, inv[2], pv[2];
        make_block(kind, A);
        hipMemcpy(dA, A.data(), NB * NB * 8, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; ++mode) {
            inv[mode].resize(NB * NB); pv[mode].resize(64);
            for (int nwg : {1, 256, 1024}) {
                double best = 1e30, mean = 0;
                for (int rep = 0; rep < 4; ++rep) {
                    hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(256), 0, 0, mode, dA, dinv, dpiv, dt);
                    hipDeviceSynchronize();
                    std::vector<unsigned long long> h(nwg);
                    hipMemcpy(h.data(), dt, nwg * 8, hipMemcpyDeviceToHost);
                    double mx = 0;
                    for (auto x : h) mx = std::fmax(mx, (double)x);
                    best = std::fmin(best, mx);
                    mean = mx;
                }
                printf("block %d  %s  %4d workgroups: %.3f us per sweep (slowest workgroup, best of 4; last %.3f)\n", kind,
                       mode ? "dpp 4x16" : "bpermute 8x8", nwg, best / 100.0 / REPS, mean / 100.0 / REPS);
            }
            hipMemcpy(inv[mode].data(), dinv, NB * NB * 8, hipMemcpyDeviceToHost);
            hipMemcpy(pv[mode].data(), dpiv, 64 * 8, hipMemcpyDeviceToHost);
        }
        const bool same = !memcmp(inv[0].data(), inv[1].data(), NB * NB * 8) && !memcmp(pv[0].data(), pv[1].data(), 64 * 8);
        double res = 0;                                 // |A X - I|_max of the new form
        for (int i = 0; i < NB; ++i)
            for (int j = 0; j < NB; ++j) {
                double acc = 0;
                for (int k = 0; k < NB; ++k) acc += A[i * NB + k] * inv[1][k * NB + j];
                res = std::fmax(res, std::fabs(acc - (i == j)));
            }
        printf("block %d: inverse + pivots bit-identical: %s   residual |A X - I|_max = %.3e\n", kind, same ? "YES" : "NO", res);
        bad |= !same;
    }
    return bad;
}
