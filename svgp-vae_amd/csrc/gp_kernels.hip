// Sparse-GP (Hensman) block of the SVGPVAE step for gfx950: kernel matrices, weighted
// statistics, m x m factor stage, per-sample stage - forward and hand-derived reverse.
//
// Reference semantics: SVGPVAE_model.py:220-343 (mainSVGP.variational_loss Hensman branch,
// approximate_posterior_params), :427-476 (mnistSVGP.kernel_matrix), :880-902 (assembly),
// utils.py:483-504 (gauss_cross_entropy).  Math and the reverse formulas: DESIGN.md section 4,
// checked stage by stage against oracle/staged_gp.py.
//
// The reference materialises a (b,m,m) tensor and several b x b products per channel
// (SVGPVAE_model.py:284-294,336-337); here everything is O(L b m^2 + L m^3) and the only
// cross-row reductions are the statistics blocks statA / statB (what data parallelism all-reduces).
#include <cstdlib>

#include "common.hpp"
#include "vae_dev.hpp"
#include "sweep32.hpp"

namespace {

// =============================================================================================
// LDS matrix helpers.  LDS matrices are m x m with leading dimension ld = m + 1 (odd-ish pad:
// column walks hit distinct banks with 8-byte elements).  All helpers are workgroup-cooperative
// and do NOT end with a barrier unless stated.
// =============================================================================================
// global -> LDS matrix copy with 4 loads in flight per thread (m = 32: the whole matrix in ONE batch).  A plain
// `R[..] = g[o]` loop compiles to load / wait / store per trip; these kernels run one workgroup of 4 waves per CU, so every
// trip would expose a full memory latency.
__device__ __forceinline__ void mat_load_nt(real* R, int ld, const real* __restrict__ g, int m, int nt) {
    const int mm = m * m;
    for (int o0 = threadIdx.x; o0 < mm; o0 += 4 * nt) {
        real v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int o = o0 + u * nt; v[u] = o < mm ? g[o] : real(0); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int o = o0 + u * nt; if (o < mm) R[(o / m) * ld + (o % m)] = v[u]; }
    }
}
__device__ __forceinline__ void mat_load(real* R, int ld, const real* __restrict__ g, int m) {
    const int mm = m * m, nt = blockDim.x;
    for (int o0 = threadIdx.x; o0 < mm; o0 += 4 * nt) {
        real v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int o = o0 + u * nt; v[u] = o < mm ? g[o] : real(0); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int o = o0 + u * nt; if (o < mm) R[(o / m) * ld + (o % m)] = v[u]; }
    }
}
__device__ __forceinline__ void mat_store(real* __restrict__ g, const real* R, int ld, int m) {
    for (int o = threadIdx.x; o < m * m; o += blockDim.x) g[o] = R[(o / m) * ld + (o % m)];
}
// Payload of an intra-launch hand-off between workgroups on different XCDs (svgp_mnist_encoder_bwd_km_sum): written THROUGH to memory
// and read past the non-coherent per-XCD L2s (sc1 = relaxed agent-scope atomics), no cache-wide write-back / invalidate -- the
// release / acquire FENCE form of the same hand-off cost the launch 10 us (every fence writes back or drops the whole L2 of its XCD
// while the image workgroups stream their partials through it).
template <bool COH> __device__ __forceinline__ real ld_co(const real* p) {
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH> __device__ __forceinline__ void st_co(real* p, real v) {
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
// ---- float64 MFMA GEMMs on LDS matrices -------------------------------------------------------
// The factor kernels keep their matrices PADDED: mp = m rounded up to 16, leading dimension
// ld = mp + 2 (conflict-free 16-row x 4-k operand fetch with 8-byte elements), pad region zero.
// One wave computes one 16x16 output tile with v_mfma_f64_16x16x4_f64 (A[i=lane&15][k=lane>>4],
// B[k=lane>>4][j=lane&15], D: col = lane&15, row = (lane>>4) + 4*reg), k-steps of 4 over mp.
// Zero pads contribute nothing and pad outputs are written as zeros, so padding is preserved.
typedef double d4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int pad16(int m) { return (m + 15) & ~15; }

template <bool TA, bool TB>
__device__ __forceinline__ d4_t mfma_tile(const real* A, const real* B, int ld, int mp, int ti, int tj) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    d4_t acc = {0, 0, 0, 0};
    const real* ap = TA ? A + q * ld + ti * 16 + r : A + (ti * 16 + r) * ld + q;
    const real* bp = TB ? B + (tj * 16 + r) * ld + q : B + q * ld + tj * 16 + r;
    const int as = TA ? 4 * ld : 4, bs = TB ? 4 : 4 * ld;
    // mp is a multiple of 16: four k-steps per trip, operands of the trip fetched together
    for (int k0 = 0; k0 < mp; k0 += 16) {
        const real a0 = ap[0], a1 = ap[as], a2 = ap[2 * as], a3 = ap[3 * as];
        const real b0 = bp[0], b1 = bp[bs], b2 = bp[2 * bs], b3 = bp[3 * bs];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc, 0, 0, 0);
        ap += 4 * as;
        bp += 4 * bs;
    }
    return acc;
}
// A matrix of <= 4 elements per thread (m * m <= 4 blockDim.x) fetched into registers now and put into LDS later: a global load in
// the MIDDLE of a dependent chain of LDS products exposes a full L2 round trip; issued at the top of the workgroup it is free
struct Mat4 { real v[4]; };
__device__ __forceinline__ Mat4 mat_fetch4(const real* __restrict__ g, int m) {
    Mat4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int o = threadIdx.x + k * blockDim.x; r.v[k] = o < m * m ? g[o] : real(0); }
    return r;
}
__device__ __forceinline__ void mat_put4(real* R, int ld, const Mat4& r, int m) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int o = threadIdx.x + k * blockDim.x; if (o < m * m) R[(o / m) * ld + (o % m)] = r.v[k]; }
}
// C = alpha * op(A) * op(B)   (all LDS padded, C must not alias A or B)
template <bool TA, bool TB>
__device__ __forceinline__ void mat_gemm(real* C, const real* A, const real* B, int ld, int m, real alpha) {
    const int mp = pad16(m), nt = mp >> 4, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    for (int t = threadIdx.x >> 6; t < nt * nt; t += (blockDim.x >> 6)) {
        const int ti = t / nt, tj = t % nt;
        const d4_t acc = mfma_tile<TA, TB>(A, B, ld, mp, ti, tj);
#pragma unroll
        for (int g = 0; g < 4; ++g) C[(ti * 16 + q + 4 * g) * ld + tj * 16 + r] = alpha * acc[g];
    }
}
// C (LDS) = alpha * op(A) * op(B) + C, tile by tile: every lane reads and writes its own elements (no barrier needed against an
// earlier mat_gemm into C; one IS needed against an element-wise pass over C)
template <bool TA, bool TB>
__device__ __forceinline__ void mat_gemm_acc(real* C, const real* A, const real* B, int ld, int m, real alpha) {
    const int mp = pad16(m), nt = mp >> 4, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    for (int t = threadIdx.x >> 6; t < nt * nt; t += (blockDim.x >> 6)) {
        const int ti = t / nt, tj = t % nt;
        const d4_t acc = mfma_tile<TA, TB>(A, B, ld, mp, ti, tj);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int idx = (ti * 16 + q + 4 * g) * ld + tj * 16 + r;
            C[idx] = alpha * acc[g] + real(1) * C[idx];
        }
    }
}
// Cg (global, m x m, ld = m) = alpha * op(A) * op(B) + C (LDS)
template <bool TA, bool TB>
__device__ __forceinline__ void mat_gemm_g_from(real* __restrict__ Cg, const real* C, const real* A, const real* B, int ld, int m,
                                                real alpha) {
    const int mp = pad16(m), nt = mp >> 4, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    for (int t = threadIdx.x >> 6; t < nt * nt; t += (blockDim.x >> 6)) {
        const int ti = t / nt, tj = t % nt;
        const d4_t acc = mfma_tile<TA, TB>(A, B, ld, mp, ti, tj);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int i = ti * 16 + q + 4 * g, j = tj * 16 + r;
            if (i < m && j < m) Cg[(size_t)i * m + j] = alpha * acc[g] + real(1) * C[i * ld + j];
        }
    }
}
// Cg (global, m x m, ld = m) = alpha * op(A) * op(B) + beta * Cg
template <bool TA, bool TB>
__device__ __forceinline__ void mat_gemm_g(real* __restrict__ Cg, const real* A, const real* B, int ld, int m,
                                           real alpha, real beta) {
    const int mp = pad16(m), nt = mp >> 4, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    for (int t = threadIdx.x >> 6; t < nt * nt; t += (blockDim.x >> 6)) {
        const int ti = t / nt, tj = t % nt;
        const d4_t acc = mfma_tile<TA, TB>(A, B, ld, mp, ti, tj);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int i = ti * 16 + q + 4 * g, j = tj * 16 + r;
            if (i < m && j < m) {
                const size_t o = (size_t)i * m + j;
                Cg[o] = alpha * acc[g] + (beta != real(0) ? beta * Cg[o] : real(0));
            }
        }
    }
}
// y = alpha * A x (+ y0), threads < m; x, y in LDS (y must not alias x)
__device__ __forceinline__ void mat_vec(real* y, const real* A, int ld, const real* x, int m, real alpha) {
    if (threadIdx.x < m) {
        real acc = 0;
#pragma unroll 8
        for (int j = 0; j < m; ++j) acc += A[threadIdx.x * ld + j] * x[j];
        y[threadIdx.x] = alpha * acc;
    }
}

// A (SPD, LDS) <- A^{-1}; returns log det A (all threads).  Gauss-Jordan elimination without
// pivoting (safe for the jittered SPD matrices of this path; pivots = the LDL^T diagonal, so
// log det = sum log pivot).  Every thread keeps its EPT elements of A in REGISTERS for all m
// steps; only the pivot row / column travel through a ping-pong LDS buffer (W, >= 5m reals), so
// a step costs one barrier.  Begins and ends with a barrier.
template <int EPT>
__device__ __forceinline__ real spd_inv_t(real* A, real* W, int ld, int m) {
    real a[EPT];
    int ii[EPT], jj[EPT];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        const int o = threadIdx.x + t * blockDim.x;
        if (o < m * m) { ii[t] = o / m; jj[t] = o % m; a[t] = A[ii[t] * ld + jj[t]]; }
        else { ii[t] = -1; jj[t] = -1; a[t] = 0; }
    }
    real logdet = 0;
    for (int k = 0; k < m; ++k) {
        real* rowk = W + (k & 1) * 2 * m;
        real* colk = rowk + m;
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            if (ii[t] == k) rowk[jj[t]] = a[t];
            if (jj[t] == k) colk[ii[t]] = a[t];
        }
        __syncthreads();
        const real piv = rowk[k];
        const real ipiv = real(1) / piv;
        if (threadIdx.x == 0) W[4 * m + k] = piv;       // logs are taken in parallel after the sweep
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            if (ii[t] >= 0) {
                const int i = ii[t], j = jj[t];
                const real rkj = (j == k) ? ipiv : rowk[j] * ipiv;
                if (i == k) a[t] = rkj;
                else a[t] = ((j == k) ? real(0) : a[t]) - colk[i] * rkj;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < EPT; ++t)
        if (ii[t] >= 0) A[ii[t] * ld + jj[t]] = a[t];
    __syncthreads();
    // log det = sum_k log(pivot_k): one log per lane of wave 0, fixed-order combine, broadcast via LDS
    if (threadIdx.x < 64) {
        real lg = 0;
        for (int k = threadIdx.x; k < m; k += 64) lg += log(W[4 * m + k]);
        lg = wave_sum(lg);
        if (threadIdx.x == 0) W[0] = lg;
    }
    __syncthreads();
    logdet = W[0];
    __syncthreads();
    return logdet;
}
// Single-wave variant for m <= MP (MP = 16 or 32): wave 0 holds the whole matrix in registers as an
// 8 x 8 grid of lanes, each owning a BS x BS block (BS = MP/8).  The k loop is fully unrolled so every
// register index is static; per step a lane needs only the BS pivot-row and BS pivot-column values of
// its block, fetched with cross-lane shuffles (ds_bpermute: register to register, no LDS memory, no
// barrier, and - unlike an LDS exchange - not subject to single-thread store/load reordering).
// Rows/columns >= m are an identity pad (pivot 1, log 0).  W: >= 1 real.  Other waves wait at the end.
__device__ __forceinline__ real fast_rcp(real x) {
    real r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, real(1)), r, r);
    r = fma(fma(-x, r, real(1)), r, r);
    return r;
}
// value of lane `src` (compile-time constant) in every lane: two v_readlane_b32 (scalar path, a few cycles) where __shfl
// issues two ds_bpermute_b32 (~50 cycles through the LDS crossbar) -- the pivot sits at the head of every step's chain
__device__ __forceinline__ real readlane_f64(real v, int src) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, src), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
template <int MP>
__device__ __forceinline__ real spd_inv_wave(real* A, real* W, int ld, int m) {
    constexpr int BS = MP / 8;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x, bi = lane >> 3, bj = lane & 7;
        real a[BS][BS];
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
            for (int c = 0; c < BS; ++c) {
                const int i = bi * BS + r, j = bj * BS + c;
                a[r][c] = (i < m && j < m) ? A[i * ld + j] : (i == j ? real(1) : real(0));
            }
        real mypiv = 1;
#pragma unroll
        for (int k = 0; k < MP; ++k) {
            if (k < m) {                                   // wave-uniform
                const int kb = k / BS, kr = k % BS;        // static after unrolling
                real rowk[BS], colk[BS];
#pragma unroll
                for (int c = 0; c < BS; ++c) rowk[c] = __shfl(a[kr][c], kb * 8 + bj, 64);
#pragma unroll
                for (int r = 0; r < BS; ++r) colk[r] = __shfl(a[r][kr], bi * 8 + kb, 64);
                const real piv = readlane_f64(a[kr][kr], kb * 9);      // uniform source lane: v_readlane, not ds_bpermute
                const real ipiv = fast_rcp(piv);
                if (lane == k) mypiv = piv;
                real rkj[BS];
#pragma unroll
                for (int c = 0; c < BS; ++c) rkj[c] = (bj * BS + c == k) ? ipiv : rowk[c] * ipiv;
#pragma unroll
                for (int r = 0; r < BS; ++r)
#pragma unroll
                    for (int c = 0; c < BS; ++c) {
                        const bool ik = (bi * BS + r == k), jk = (bj * BS + c == k);
                        a[r][c] = ik ? rkj[c] : ((jk ? real(0) : a[r][c]) - colk[r] * rkj[c]);
                    }
            }
        }
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
            for (int c = 0; c < BS; ++c) {
                const int i = bi * BS + r, j = bj * BS + c;
                if (i < m && j < m) A[i * ld + j] = a[r][c];
            }
        const real lg = wave_sum(log(mypiv));
        if (lane == 0) W[0] = lg;
    }
    __syncthreads();
    const real logdet = W[0];
    __syncthreads();
    return logdet;
}

// 16 < m <= 32: the 4 x 16-lane form of sweep32.hpp (pivot column by DPP row broadcast; same bits as spd_inv_wave<32>, 1.4x
// faster: 54 instead of 78 instructions per pivot on a wave that issues one per ~5.8 cycles).  Other waves wait at the end.
__device__ __forceinline__ real spd_inv_wave32_dpp(real* A, real* W, int ld, int m) {
    __syncthreads();
    if (threadIdx.x < 64) {
        const real mypiv = sweep32::gauss_jordan_32(
            threadIdx.x, m, [&](int i, int j) { return (i < m && j < m) ? A[i * ld + j] : (i == j ? real(1) : real(0)); },
            [&](int i, int j, real v) {
                if (i < m && j < m) A[i * ld + j] = v;
            });
        const real lg = wave_sum(log(mypiv));
        if (threadIdx.x == 0) W[0] = lg;
    }
    __syncthreads();
    const real logdet = W[0];
    __syncthreads();
    return logdet;
}

// W must hold >= max(5 m, 65) reals.
__device__ __forceinline__ real chol_inv(real* A, real* W, int ld, int m) {
    if (m <= 16) return spd_inv_wave<16>(A, W, ld, m);
    if (m <= 32) return spd_inv_wave32_dpp(A, W, ld, m);
    return spd_inv_t<16>(A, W, ld, m);       // SVGP_BLOCK threads, m <= 64 -> 16 elements per thread
}

// =============================================================================================
// kernel matrices  (SVGPVAE_model.py:427-476; TFP ExpSinSquared x Linear)
// =============================================================================================
typedef SvgpKernArgs KernArgs;
// the batch-row side of the VJP keeps the m x M inducing object vectors in LDS while they fit in 96 KB (config 3: 64 KB)
__host__ __device__ __forceinline__ bool km_stage_O(int m, int M) { return (size_t)m * M * sizeof(real) <= 96 * 1024; }
__device__ __forceinline__ const real* obj_row(const KernArgs& a, int n) { return svgp_obj_row(a, n); }
__device__ __forceinline__ real view_k(real d, real a2, real inv_l2) { return svgp_view_k(d, a2, inv_l2); }
__device__ __forceinline__ real dotM(const real* x, const real* y, int M) { return svgp_dotM(x, y, M); }

__global__ __launch_bounds__(SVGP_BLOCK) void k_kernel_matrix_fwd(KernArgs a, real* __restrict__ K,
                                                                  real* __restrict__ Kn, real* __restrict__ knn) {
    svgp_km_fwd_element(a, (long long)blockIdx.x * blockDim.x + threadIdx.x, K, Kn, knn);
}

// General argument patterns of mnistSVGP.kernel_matrix (SVGPVAE_model.py:427-476): each side's object vector is either
// the row's own columns 2: (`*_inducing`, or no GPLVM table) or gathered from the table by the row's id (:451,455).
struct KernXYArgs {
    int M, normalize, nx, ny, xg, yg, diag;
    const real *x, *y, *table, *ls, *amp;
};
__global__ __launch_bounds__(SVGP_BLOCK) void k_kernel_matrix_xy(KernXYArgs a, real* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = a.diag ? (long long)a.nx : (long long)a.nx * a.ny;
    if (idx >= total) return;
    const int st = 2 + a.M;
    const long long i = a.diag ? idx : idx / a.ny, j = a.diag ? idx : idx % a.ny;
    const real* xr = a.x + (size_t)i * st;
    const real* yr = a.y + (size_t)j * st;
    const real* xo = a.xg ? a.table + (size_t)((long long)xr[0]) * a.M : xr + 2;
    const real* yo = a.yg ? a.table + (size_t)((long long)yr[0]) * a.M : yr + 2;
    const real amp = *a.amp, ls = *a.ls;
    real D = dotM(xo, yo, a.M);
    if (a.normalize) D /= sqrt(dotM(xo, xo, a.M)) * sqrt(dotM(yo, yo, a.M));
    out[idx] = view_k(xr[1] - yr[1], amp * amp, real(1) / (ls * ls)) * D;
}

// VJP, inducing side: one workgroup per inducing point j.
// d_ip[j] = [0, d_theta_j, d_o_j], partial amplitude / length-scale sums -> part_gp[j].
#define KM_MAXM 32
#define KM_MTOT 128            // GPLVM dimension limit of this build (api.hip): the feature index is walked in chunks of KM_MAXM
// MC / MM: compile-time m / GPLVM dimension M (0: run time); config 2 runs the <32, 8> instance (8 instead of 32 predicated
// accumulators per thread, index divisions folded).  M > 32 (a kernel matrix of rank m needs M >= m / 16 object
// dimensions -- m = 2048 wants M = 128): the row / column loops are repeated per chunk of 32 feature columns, 32 accumulators live.
template <int MC, int MM, int NTL = 0, bool COH = false>
__device__ __forceinline__ void km_bwd_cols(const int j, const KernArgs& a, real rep_weight, int train_ip,
                                            const real* __restrict__ K, const real* __restrict__ Kn,
                                            const real* __restrict__ Kbar, const real* __restrict__ Knbar,
                                            real* __restrict__ d_ip, real* __restrict__ part_gp) {
    __shared__ real res[KM_MTOT + 4];                      // [0, M): object-vector gradient; KM_MTOT + {0, 1, 2}: amp, ls, theta
    __shared__ real wred[SVGP_BLOCK / 64][KM_MAXM + 4];
    const int M = MM ? MM : a.M, m = MC ? MC : a.m, st = 2 + M;
    const int nthr = NTL ? NTL : (int)blockDim.x;          // live threads (NTL: a rider of a launch with more threads per workgroup)
    const real amp = *a.amp, ls = *a.ls, a2 = amp * amp, inv_l2 = real(1) / (ls * ls);
    const real* oj = a.ip + (size_t)j * st + 2;
    const real thj = a.ip[(size_t)j * st + 1];
    const real nj = a.normalize ? sqrt(dotM(oj, oj, M)) : real(1);
    for (int k0 = 0; k0 < M; k0 += KM_MAXM) {
        const int Mc = M - k0 < KM_MAXM ? M - k0 : KM_MAXM;
        real acc_amp = 0, acc_ls = 0, acc_th = 0, acc_o[KM_MAXM];
#pragma unroll
        for (int k = 0; k < KM_MAXM; ++k) acc_o[k] = 0;
        // ---- K_nm column j
        for (int n = threadIdx.x; n < a.b; n += nthr) {
            const real gk = ld_co<COH>(Knbar + (size_t)n * m + j);
            const real d = a.aux[(size_t)n * st + 1] - thj;
            const real V = view_k(d, a2, inv_l2);
            const real G = gk * Kn[(size_t)n * m + j];
            const real sh = sin(real(0.5) * d);
            acc_amp += G;
            acc_ls += G * sh * sh;
            acc_th += G * sin(d);
            const real* on = obj_row(a, n);
            const real c = gk * V / (a.normalize ? sqrt(dotM(on, on, M)) : real(1));
#pragma unroll
            for (int k = 0; k < KM_MAXM; ++k)
                if (k < Mc) acc_o[k] += c * on[k0 + k];
        }
        // ---- K_mm: row j (first argument) and column j (second argument), replicated across ranks
        for (int i = threadIdx.x; i < m; i += nthr) {
            const real* oi = a.ip + (size_t)i * st + 2;
            const real ni = a.normalize ? sqrt(dotM(oi, oi, M)) : real(1);
            const real d = thj - a.ip[(size_t)i * st + 1];          // theta_j - theta_i  (entry (j,i))
            const real V = view_k(d, a2, inv_l2);                   // even in d
            const real sh = sin(real(0.5) * d);
            const real g_ji = rep_weight * ld_co<COH>(Kbar + (size_t)j * m + i), g_ij = rep_weight * ld_co<COH>(Kbar + (size_t)i * m + j);
            const real G_ji = g_ji * K[(size_t)j * m + i], G_ij = g_ij * K[(size_t)i * m + j];
            acc_amp += G_ji;
            acc_ls += G_ji * sh * sh;
            // entry (j,i): dK/dtheta_j = -K sin(d);  entry (i,j): dK/dtheta_j = +K sin(theta_i - theta_j) = -K sin(d)
            acc_th += -(G_ji + G_ij) * sin(d);
            const real c = (g_ji + g_ij) * V / ni;
#pragma unroll
            for (int k = 0; k < KM_MAXM; ++k)
                if (k < Mc) acc_o[k] += c * oi[k0 + k];
        }
        // one combined reduction per chunk: wave sums -> LDS [wave][3+Mc] -> thread v sums the waves (fixed order)
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        __syncthreads();                                           // (wred of the previous chunk has been read)
        real t = wave_sum(acc_amp); if (lane == 0) wred[wv][KM_MAXM] = t;
        t = wave_sum(acc_ls);       if (lane == 0) wred[wv][KM_MAXM + 1] = t;
        t = wave_sum(acc_th);       if (lane == 0) wred[wv][KM_MAXM + 2] = t;
#pragma unroll
        for (int k = 0; k < KM_MAXM; ++k)
            if (k < Mc) { t = wave_sum(acc_o[k]); if (lane == 0) wred[wv][k] = t; }
        __syncthreads();
        if (threadIdx.x < KM_MAXM + 3 && (threadIdx.x < Mc || (threadIdx.x >= KM_MAXM && k0 == 0))) {
            real sres = 0;
            for (int wq = 0; wq < (nthr >> 6); ++wq) sres += wred[wq][threadIdx.x];
            res[threadIdx.x < KM_MAXM ? k0 + threadIdx.x : KM_MTOT + (threadIdx.x - KM_MAXM)] = sres;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part_gp[j * 2 + 0] = real(2) * res[KM_MTOT] / amp;                         // d amplitude
        part_gp[j * 2 + 1] = real(4) * res[KM_MTOT + 1] / (ls * ls * ls);          // d length scale
        real* out = d_ip + (size_t)j * st;
        out[0] = 0;
        out[1] = train_ip ? res[KM_MTOT + 2] * inv_l2 : real(0);
        // res[k] is the gradient w.r.t. the (normalised) o_j direction: d_oh; chain through o/|o|
        real proj = 0;
        if (a.normalize)
            for (int k = 0; k < M; ++k) proj += res[k] * oj[k] / nj;
        for (int k = 0; k < M; ++k) {
            real g = res[k];
            if (a.normalize) g = (g - proj * oj[k] / nj) / nj;
            out[2 + k] = train_ip ? g : real(0);
        }
    }
}

// VJP, batch-row side.  grid ceil(b/RB), RB = 256/m rows per workgroup.  Phase 1: thread (row, j)
// computes c = Knbar * view(theta_n - theta_j) / |o_j|; phase 2: thread (row, k) reduces over j ->
// d_on (b,M), the gradient of the gathered object row.  Also the k_nn part of the amplitude gradient.
template <int MC, int MM, int NTL = 0, bool COH = false>
__device__ __forceinline__ void km_bwd_rows(const int rblk, const KernArgs& a, const real* __restrict__ Knbar,
                                            const real* __restrict__ knnbar, const real* __restrict__ knn,
                                            real* __restrict__ d_on, real* __restrict__ part_gp) {
    extern __shared__ __align__(16) real smem[];
    __shared__ real red[16];
    const int nthr = NTL ? NTL : (int)blockDim.x;
    const int m = MC ? MC : a.m, M = MM ? MM : a.M, st = 2 + M, RB = m >= nthr ? 1 : nthr / m;
    // m x M inducing object vectors, divided by their norm when normalising: staged in LDS while they fit (km_stage_O, the
    // host sizes the launch with the same rule); else read from the parameter vector with the inverse norms (m) staged
    const bool stage_O = km_stage_O(m, M);
    real* O = smem;
    real* cbuf = O + (stage_O ? m * M : m);     // RB x m
    real* gbuf = cbuf + RB * m;                 // RB x M
    const real amp = *a.amp, ls = *a.ls, a2 = amp * amp, inv_l2 = real(1) / (ls * ls);
    if (stage_O) {
        for (int o = threadIdx.x; o < m * M; o += nthr) {
            const int j = o / M;
            const real* oj = a.ip + (size_t)j * st + 2;
            const real nj = a.normalize ? sqrt(dotM(oj, oj, M)) : real(1);
            O[o] = oj[o % M] / nj;
        }
    } else {
        for (int j = threadIdx.x; j < m; j += nthr) {
            const real* oj = a.ip + (size_t)j * st + 2;
            O[j] = a.normalize ? real(1) / sqrt(dotM(oj, oj, M)) : real(1);
        }
    }
    for (int it = threadIdx.x; it < RB * m; it += nthr) {
        const int nl = it / m, j = it % m, n = rblk * RB + nl;
        real c = 0;
        if (n < a.b)
            c = ld_co<COH>(Knbar + (size_t)n * m + j) * view_k(a.aux[(size_t)n * st + 1] - a.ip[(size_t)j * st + 1], a2, inv_l2);
        cbuf[it] = c;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < RB * M; it += nthr) {
        const int nl = it / M, k = it % M, n = rblk * RB + nl;
        real g = 0;
        if (n < a.b) {
            if (stage_O) for (int j = 0; j < m; ++j) g += cbuf[nl * m + j] * O[j * M + k];
            else for (int j = 0; j < m; ++j) g += cbuf[nl * m + j] * (a.ip[(size_t)j * st + 2 + k] * O[j]);
            const real* on = obj_row(a, n);
            const real nn = a.normalize ? sqrt(dotM(on, on, M)) : real(1);
            g += real(2) * a2 * ld_co<COH>(knnbar + n) * on[k] / nn;      // k_nn = a^2 |o_hat|^2
        }
        gbuf[it] = g;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < RB * M; it += nthr) {
        const int nl = it / M, k = it % M, n = rblk * RB + nl;
        if (n < a.b) {
            real v = gbuf[it];
            if (a.normalize) {       // o_hat = o/|o| : d_o = (d_oh - <d_oh,o_hat> o_hat)/|o|
                const real* on = obj_row(a, n);
                const real nn = sqrt(dotM(on, on, M));
                real proj = 0;
                for (int kk = 0; kk < M; ++kk) proj += gbuf[nl * M + kk] * on[kk] / nn;
                v = (v - proj * on[k] / nn) / nn;
            }
            d_on[(size_t)n * M + k] = v;
        }
    }
    real acc_amp = 0;
    {
        const int n = rblk * RB + threadIdx.x;
        if (threadIdx.x < RB && n < a.b) acc_amp = real(2) * ld_co<COH>(knnbar + n) * knn[n] / amp;
    }
    acc_amp = block_sum_nt(acc_amp, red, nthr);
    if (threadIdx.x == 0) {
        part_gp[(m + rblk) * 2 + 0] = acc_amp;
        part_gp[(m + rblk) * 2 + 1] = 0;
    }
}

// one launch for both sides of the VJP: workgroups [0, m) = inducing side, [m, m + nrb) = batch-row side
template <int MC, int MM>
__global__ __launch_bounds__(SVGP_BLOCK) void k_kernel_matrix_bwd_cr(KernArgs a, real rep_weight, int train_ip,
                                                                     const real* __restrict__ K,
                                                                     const real* __restrict__ Kn,
                                                                     const real* __restrict__ Kbar,
                                                                     const real* __restrict__ Knbar,
                                                                     const real* __restrict__ knnbar,
                                                                     const real* __restrict__ knn,
                                                                     real* __restrict__ d_ip, real* __restrict__ d_on,
                                                                     real* __restrict__ part_gp) {
    const int m = MC ? MC : a.m;
    if ((int)blockIdx.x < m) km_bwd_cols<MC, MM>(blockIdx.x, a, rep_weight, train_ip, K, Kn, Kbar, Knbar, d_ip, part_gp);
    else km_bwd_rows<MC, MM>(blockIdx.x - m, a, Knbar, knnbar, knn, d_on, part_gp);
}

// Deterministic scatter-add of d_on into the object table gradient + the final amplitude / length-scale sums
// (svgp_km_scatter_block in common.hpp; the training phases run these workgroups inside the gradient-reduction
// launch instead, see svgp_mnist_grad_reduce_all).
__global__ __launch_bounds__(SVGP_BLOCK) void k_kernel_matrix_bwd_scatter(KernArgs a, int n_gp_part, int train_gp,
                                                                          int train_ov,
                                                                          const real* __restrict__ d_on,
                                                                          const real* __restrict__ part_gp,
                                                                          real* __restrict__ d_ov,
                                                                          real* __restrict__ d_ls,
                                                                          real* __restrict__ d_amp) {
    svgp_km_scatter_block(blockIdx.x, gridDim.x, a.b, a.M, a.n_obj, a.aux, n_gp_part, train_gp, train_ov, d_on, part_gp,
                          d_ov, d_ls, d_amp);
}

// =============================================================================================
// weighted statistics  S_l = Kn^T diag(w_l) Kn,  v1_l = Kn^T a_l,  v2_l = Kn^T b_l
// grid (T, L [+1]): blockIdx.y = channel, blockIdx.x = tile of 256 outputs of the m x m matrix.
// mode 0 (forward):  w = 1/var, a = y/var                      (SVGPVAE_model.py:328-334)
// mode 1 (backward): computes g_pv, g_pm, mvbar (stored) and uses w = g_pv, a = mvbar, b = c g_pm
// In forward mode the extra block blockIdx.y == L inverts K_mm + jitter I (:239,270,273).
// =============================================================================================
struct StatArgs {
    int b, m, L, mode, rc_rows, clip_pv;   // rc_rows: rows of K_nm staged per pass
    int with_aji;                          // backward mode: workgroups y in [L, 2L) finish (A_hat + jI)^-1 and KL
    const real* Ahat; real* Aji; real* KL; // ... of channel y - L (deferred from svgp_gp_factor_fwd, off its critical path)
    real c, jitter, beta_over_L_unused;
    int geco;
    const real* Kn;      // (b,m)
    const real* y;       // qnet_mu (b,L)
    const real* s2;      // qnet_var (b,L)
    // backward-only inputs
    const real* p_m; const real* p_v; const real* e; const real* eps; const real* zbar; const real* state;
    real* g_pv; real* g_pm; real* mvbar;
    // outputs
    real* S; real* v1; real* v2;
    // forward-only: K -> Ki, ldK
    const real* K; real* Ki; real* ldK;
};

__device__ __forceinline__ real grad_KL_term(int flags, int L, const real* state) { return svgp_seed_T(flags, L, state); }

// Statistics of channel l, row partial `part` of `nparts_grid` (the body of k_gp_stats; COH: the outputs S, v1, v2 are written
// through for a consumer in the SAME launch, svgp_gp_stats_factor_bwd_wgrad)
template <int MC, bool COH>
__device__ __forceinline__ void gp_stats_block(const StatArgs& a, int part, int nparts_grid, int l, real* smem) {
    const int m = MC ? MC : a.m;
    // ---- statistics of channel l: S = Kn^T diag(w) Kn on the f64 MFMA (A[i][k=n] = w_n Kn[n][i],
    // B[k=n][j] = Kn[n][j], k-steps of 4 rows), vectors on the VALU.  One wave per 16x16 tile.
    const int STAT_RC = a.rc_rows;            // multiple of 4
    const int mp = pad16(m), ldk = mp + 2, nt = mp >> 4;
    real* kt = smem;                   // STAT_RC x ldk tile of Kn, zero padded
    real* w = kt + STAT_RC * ldk;      // STAT_RC
    real* va = w + STAT_RC;            // STAT_RC
    real* vb = va + STAT_RC;           // STAT_RC
    real* scr = vb + STAT_RC;          // 2 x SVGP_BLOCK
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q = lane >> 4;
    const int vi = threadIdx.x % mp, vpart = threadIdx.x / mp, nparts = blockDim.x / mp;
    d4_t acc[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[tt] = d4_t{0, 0, 0, 0};
    real acc1 = 0, acc2 = 0;
    const real gT = a.mode ? grad_KL_term(a.geco, a.L, a.state) : real(0);
    // row partial blockIdx.x of gridDim.x: rows [rlo, rhi) -> partial block `part` of the outputs
    const int RP = ((a.b + nparts_grid - 1) / nparts_grid + 3) & ~3;
    const int rlo = min(a.b, part * RP), rhi = min(a.b, rlo + RP);
    const size_t pl_ = (size_t)part * a.L + l;
    for (int r0 = rlo; r0 < rhi; r0 += STAT_RC) {
        const int rows = min(STAT_RC, rhi - r0), rows4 = (rows + 3) & ~3;
        __syncthreads();
        if (vpart < nparts) {          // thread = (column vi, row lane vpart): no division in the loop
#pragma unroll 4
            for (int rr = vpart; rr < rows4; rr += nparts)
                kt[rr * ldk + vi] = (rr < rows && vi < m) ? a.Kn[(size_t)(r0 + rr) * m + vi] : real(0);
        }
        for (int rr = threadIdx.x; rr < rows4; rr += blockDim.x) {
            if (rr >= rows) { w[rr] = 0; va[rr] = 0; vb[rr] = 0; continue; }
            const size_t e = (size_t)(r0 + rr) * a.L + l;
            const real p = recip_no_nan(a.s2[e]);
            if (a.mode == 0) {
                w[rr] = p;
                va[rr] = p * a.y[e];
                vb[rr] = 0;
            } else {
                const real zb = a.zbar[e];
                const real pv = a.p_v[e];
                real gpv = svgp_gpv(a.clip_pv, gT, p, zb, a.eps[e], pv);
                const real gpm = gT * p * (a.p_m[e] - a.y[e]) + zb;
                const real mvb = svgp_seed_3(a.geco, gT) * p * a.e[e];
                w[rr] = gpv;
                va[rr] = mvb;
                vb[rr] = a.c * gpm;
                a.g_pv[e] = gpv; a.g_pm[e] = gpm; a.mvbar[e] = mvb;
            }
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int t = wave + 4 * tt;
            if (t < nt * nt) {
                const int ti = t / nt, tj = t % nt;
                const real* ap = kt + q * ldk + ti * 16 + r16;
                const real* bp = kt + q * ldk + tj * 16 + r16;
                int k0 = 0;
                for (; k0 + 16 <= rows4; k0 += 16) {          // four k-steps per trip, loads issued together
                    const real a0 = ap[0], a1 = ap[4 * ldk], a2 = ap[8 * ldk], a3 = ap[12 * ldk];
                    const real b0 = bp[0], b1 = bp[4 * ldk], b2 = bp[8 * ldk], b3 = bp[12 * ldk];
                    const real w0 = w[k0 + q], w1 = w[k0 + 4 + q], w2 = w[k0 + 8 + q], w3 = w[k0 + 12 + q];
                    acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0 * w0, b0, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1 * w1, b1, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2 * w2, b2, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3 * w3, b3, acc[tt], 0, 0, 0);
                    ap += 16 * ldk;
                    bp += 16 * ldk;
                }
                for (; k0 < rows4; k0 += 4) {
                    acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[0] * w[k0 + q], bp[0], acc[tt], 0, 0, 0);
                    ap += 4 * ldk;
                    bp += 4 * ldk;
                }
            }
        }
        if (vpart < nparts && vi < m) {
#pragma unroll 8
            for (int r = vpart; r < rows; r += nparts) {
                const real k = kt[r * ldk + vi];
                acc1 += va[r] * k;
                acc2 += vb[r] * k;
            }
        }
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        const int t = wave + 4 * tt;
        if (t < nt * nt) {
            const int ti = t / nt, tj = t % nt;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = ti * 16 + q + 4 * g, j = tj * 16 + r16;
                if (i < m && j < m) st_co<COH>(a.S + pl_ * m * m + (size_t)i * m + j, acc[tt][g]);
            }
        }
    }
    __syncthreads();
    scr[threadIdx.x] = acc1;
    scr[SVGP_BLOCK + threadIdx.x] = acc2;
    __syncthreads();
    if (threadIdx.x < m) {
        real s1 = 0, s2 = 0;
        for (int pp = 0; pp < nparts; ++pp) { s1 += scr[pp * mp + threadIdx.x]; s2 += scr[SVGP_BLOCK + pp * mp + threadIdx.x]; }
        st_co<COH>(a.v1 + pl_ * m + threadIdx.x, s1);
        if (a.v2) st_co<COH>(a.v2 + pl_ * m + threadIdx.x, s2);
    }
}

template <int MC>
__global__ __launch_bounds__(SVGP_BLOCK) void k_gp_stats(StatArgs a) {
    extern __shared__ __align__(16) real smem[];
    const int m = MC ? MC : a.m, l = blockIdx.y;
    if (a.mode == 0 && l == a.L) {   // ---- K_mm inverse block (forward mode only, blockIdx.x == 0)
        if (blockIdx.x != 0) return;
        const int ld = m + 1;
        real* A = smem;
        real* W = A + m * ld;
        mat_load(A, ld, a.K, m);
        __syncthreads();
        if (threadIdx.x < m) A[threadIdx.x * ld + threadIdx.x] += a.jitter;
        const real logdet = chol_inv(A, W, ld, m);
        mat_store(a.Ki, A, ld, m);
        if (threadIdx.x == 0) *a.ldK = logdet;
        return;
    }
    if (a.mode == 1 && (int)blockIdx.y >= a.L) {
        // ---- deferred tail of the forward factor stage: Aji = (A_hat + jitter I)^-1 and the log det term of KL
        // (SVGPVAE_model.py:271-279).  Only the reverse pass and the KL scalar need them, so they run here, beside
        // the L statistics workgroups that leave most CUs idle, instead of on the forward critical path.
        if (blockIdx.x != 0) return;
        const int l2 = blockIdx.y - a.L, ld = m + 1;
        real* A = smem;
        real* W = A + m * ld;
        mat_load(A, ld, a.Ahat + (size_t)l2 * m * m, m);
        __syncthreads();
        if (threadIdx.x < m) A[threadIdx.x * ld + threadIdx.x] += a.jitter;
        const real ldA = chol_inv(A, W, ld, m);
        mat_store(a.Aji + (size_t)l2 * m * m, A, ld, m);
        if (threadIdx.x == 0) a.KL[l2] -= real(0.5) * ldA;
        return;
    }
    gp_stats_block<MC, false>(a, (int)blockIdx.x, (int)gridDim.x, l, smem);
}

// =============================================================================================
// m x m factor stage, forward.  blockIdx.x < L: channel l.  blockIdx.x >= L: q_n = k_n^T Ki k_n
// for a block of rows (shared by all channels).
// =============================================================================================
// sum of the row partials of one statistics element (stride between partial blocks); P is 1 or SVGP_STAT_PARTS, the
// loads of the unrolled form are independent
template <bool COH = false>
__device__ __forceinline__ real part_sum(const real* p, size_t stride, int P) {
    if (P == 1) return ld_co<COH>(p);
    real s[SVGP_STAT_PARTS];
#pragma unroll
    for (int pp = 0; pp < SVGP_STAT_PARTS; ++pp) s[pp] = ld_co<COH>(p + pp * stride);
    real t = s[0];
#pragma unroll
    for (int pp = 1; pp < SVGP_STAT_PARTS; ++pp) t += s[pp];
    return t;
}

struct FactArgs {
    int b, m, L;
    int defer_aji;           // 1: (A_hat + jI)^-1 and its log det are finished by svgp_gp_stats_bwd's extra workgroups
    int kl_form;             // cfg.kl_form
    int P;                   // row partials of S, v
    real c, jitter;
    const real* K; const real* Ki; const real* ldK; const real* S; const real* v; const real* Kn;
    real* Si; real* t; real* G; real* A; real* Aji; real* mu; real* u; real* M2; real* KL; real* q;
};

// MC: compile-time m (0: a.m at run time).  With m = 32 known (config 2) the index arithmetic of every element loop (o / m,
// o % m with a run-time divisor: ~35 instructions each) folds to shifts and the loops unroll.
template <int MC>
__global__ __launch_bounds__(SVGP_BLOCK) void k_gp_factor_fwd(FactArgs a) {
    extern __shared__ __align__(16) real smem[];
    const int m = MC ? MC : a.m, mp = pad16(m), ld = mp + 2, mm = mp * ld;
    real* R0 = smem;
    real* R1 = R0 + mm;
    real* R2 = R1 + mm;
    real* R3 = R2 + mm;
    for (int o = threadIdx.x; o < 4 * mm; o += blockDim.x) smem[o] = 0;   // zero pads (MFMA tiles read them)
    __syncthreads();
    real* vx = R3 + mm;       // m
    real* vy = vx + m;        // m
    real* vz = vy + m;        // m
    real* red = vz + m;       // 16 (+ row scratch 256 for the q blocks)
    if ((int)blockIdx.x >= a.L) {
        // ---- q rows: thread (n_local, i); rows per block RB = blockDim / m
        const int RB = blockDim.x / m, rb = blockIdx.x - a.L;
        const int nl = threadIdx.x / m, i = threadIdx.x % m, n = rb * RB + nl;
        const bool act = nl < RB && n < a.b;
        mat_load(R0, ld, a.Ki, m);
        real* kr = R1;          // RB x m rows of Kn
        if (act) kr[nl * m + i] = a.Kn[(size_t)n * m + i];
        __syncthreads();
        real acc = 0;
        if (act) {
#pragma unroll 8
            for (int j = 0; j < m; ++j) acc += R0[j * ld + i] * kr[nl * m + j];   // Ki symmetric
            acc *= kr[nl * m + i];
        }
        real* sc = red + 16;
        sc[threadIdx.x] = act ? acc : real(0);
        __syncthreads();
        if (act && i == 0) {
            real s = 0;
            for (int k = 0; k < m; ++k) s += sc[nl * m + k];
            a.q[n] = s;
        }
        return;
    }
    const int l = blockIdx.x;
    const size_t om = (size_t)l * m * m, ov = (size_t)l * m;
    // (m * m <= 4 * 256: Ki, needed in the middle of the chain, is fetched into registers now)
    const bool keep = m * m <= 4 * (int)blockDim.x;
    Mat4 pKi = {};
    if (keep) pKi = mat_fetch4(a.Ki, m);
    // R0 = K ; R1 = K + c S_l + jitter I
    mat_load(R0, ld, a.K, m);
    for (int o = threadIdx.x; o < m * m; o += blockDim.x) {
        const int i = o / m, j = o % m;
        R1[i * ld + j] = a.K[o] + a.c * part_sum(a.S + om + o, (size_t)a.L * m * m, a.P) + (i == j ? a.jitter : real(0));
    }
    if (threadIdx.x < m) vx[threadIdx.x] = part_sum(a.v + ov + threadIdx.x, (size_t)a.L * m, a.P);
    chol_inv(R1, R2, ld, m);                       // R1 = Sigma_l^{-1}
    mat_store(a.Si + om, R1, ld, m);
    mat_vec(vy, R1, ld, vx, m, real(1));           // t = Si v
    mat_gemm<false, false>(R2, R1, R0, ld, m, real(1));   // G = Si K
    __syncthreads();
    if (threadIdx.x < m) a.t[ov + threadIdx.x] = vy[threadIdx.x];
    mat_store(a.G + om, R2, ld, m);
    mat_gemm<false, false>(R3, R0, R2, ld, m, real(1));   // A = K G
    mat_vec(vz, R0, ld, vy, m, a.c);               // mu_hat = c K t
    __syncthreads();
    mat_store(a.A + om, R3, ld, m);
    if (threadIdx.x < m) a.mu[ov + threadIdx.x] = vz[threadIdx.x];
    if (keep) mat_put4(R1, ld, pKi, m); else mat_load(R1, ld, a.Ki, m);      // R1 = Ki   (Si, G no longer needed in LDS)
    __syncthreads();
    mat_vec(vx, R1, ld, vz, m, real(1));           // u = Ki mu_hat
    // tr(Ki A) = sum_ij Ki_ij A_ji
    real tr = 0;
    for (int o = threadIdx.x; o < m * m; o += blockDim.x) tr += R1[(o / m) * ld + (o % m)] * R3[(o % m) * ld + (o / m)];
    mat_gemm<false, false>(R2, R3, R1, ld, m, real(1));   // T = A Ki
    __syncthreads();
    tr = block_sum(tr, red);
    real muu = 0;
    if (threadIdx.x < m) {
        a.u[ov + threadIdx.x] = vx[threadIdx.x];
        muu = vz[threadIdx.x] * vx[threadIdx.x];
    }
    muu = block_sum(muu, red);
    if (a.kl_form) {
        // moving-ball KL (SVGPVAE_model.py:135-137): tr(Ki A A) = sum_ij (A Ki)_ij A_ji in the place of mu.u
        real qq = 0;
        for (int o = threadIdx.x; o < m * m; o += blockDim.x) qq += R2[(o / m) * ld + (o % m)] * R3[(o % m) * ld + (o / m)];
        qq = block_sum(qq, red);
        if (threadIdx.x == 0) a.KL[a.L + l] = qq;
        muu = (real)a.L * qq;
    }
    mat_gemm<false, false>(R0, R1, R2, ld, m, real(1));   // M2 = Ki A Ki   (K no longer needed)
    __syncthreads();
    mat_store(a.M2 + om, R0, ld, m);
    if (a.defer_aji) {
        if (threadIdx.x == 0) a.KL[l] = real(0.5) * (*a.ldK - (real)m + tr + muu);     // - ldA / 2 follows
        return;
    }
    if (threadIdx.x < m) R3[threadIdx.x * ld + threadIdx.x] += a.jitter;     // A + jitter I
    const real ldA = chol_inv(R3, R2, ld, m);
    mat_store(a.Aji + om, R3, ld, m);
    if (threadIdx.x == 0) a.KL[l] = real(0.5) * (*a.ldK - ldA - (real)m + tr + muu);
}

// =============================================================================================
// per-sample stage, forward.  grid (ceil(b/RB), L); thread (n_local, i), RB = 256 / m rows.
// =============================================================================================
struct PostArgs {
    int b, m, L, clip_pv;
    real c;
    int use_rng;
    const real* Kn; const real* knn; const real* q; const real* y; const real* s2;
    const real* Si; const real* M2; const real* t; const real* u;
    const real* eps_in; const real* state;
    real* eps; real* p_m; real* p_v; real* e; real* d; real* z;
    real* part;     // (L * nb, 2) partial [L3 data term, CE]
    int nb;         // row blocks (grid x; one more column of workgroups when with_aji)
    // with_aji: workgroup (nb, l) finishes (A_hat_l + jI)^-1 and the log det term of KL_l (deferred by svgp_gp_factor_fwd_defer_aji)
    int with_aji; real jitter; const real* Ahat; real* Aji; real* KL;
};

__device__ __forceinline__ real philox_normal(unsigned long long ctr, unsigned long long idx) { return svgp_philox_normal(ctr, idx); }

template <int MC>
__global__ __launch_bounds__(SVGP_BLOCK) void k_gp_posterior_fwd(PostArgs a) {
    extern __shared__ __align__(16) real smem[];
    const int m = MC ? MC : a.m, ld = m + 1, mm = m * ld, l = blockIdx.y;
    const int bx = (int)blockIdx.x - a.with_aji;      // the riders take x = 0: first in dispatch order, they run longest
    if (bx < 0) {
        // ---- deferred tail of the forward factor stage: Aji = (A_hat + jitter I)^-1 and the log det term of KL
        // (SVGPVAE_model.py:271-279).  Only the reverse pass and the KL scalar need them.  This launch has several
        // workgroups per CU in flight, so the 8.8 us single-wave sweep runs beside the row blocks instead of bounding
        // the reverse statistics launch (which needs 5.7 us without it; riding in the decoder launch does not work:
        // its 256 image workgroups fill the chip one per CU and a rider would wait for the first of them to finish).
        real* A = smem;
        real* W = A + m * ld;
        mat_load(A, ld, a.Ahat + (size_t)l * m * m, m);
        __syncthreads();
        if ((int)threadIdx.x < m) A[threadIdx.x * ld + threadIdx.x] += a.jitter;
        const real ldA = chol_inv(A, W, ld, m);
        mat_store(a.Aji + (size_t)l * m * m, A, ld, m);
        if (threadIdx.x == 0) a.KL[l] -= real(0.5) * ldA;
        return;
    }
    real* R0 = smem;            // Si_l
    real* R1 = R0 + mm;         // M2_l
    real* tv = R1 + mm;         // t_l
    real* uv = tv + m;          // u_l
    real* kr = uv + m;          // RB x m
    real* sc = kr + SVGP_BLOCK; // 4 x 256 partial products
    real* red = sc + 4 * SVGP_BLOCK;
    const int RB = blockDim.x / m;
    const int nl = threadIdx.x / m, i = threadIdx.x % m, n = bx * RB + nl;
    const bool act = nl < RB && n < a.b;
    const size_t om = (size_t)l * m * m;
    mat_load(R0, ld, a.Si + om, m);
    mat_load(R1, ld, a.M2 + om, m);
    if (threadIdx.x < m) { tv[threadIdx.x] = a.t[(size_t)l * m + threadIdx.x]; uv[threadIdx.x] = a.u[(size_t)l * m + threadIdx.x]; }
    if (act) kr[nl * m + i] = a.Kn[(size_t)n * m + i];
    __syncthreads();
    real r = 0, s = 0, pm = 0, mv = 0;
    if (act) {
        const real ki = kr[nl * m + i];
#pragma unroll 8
        for (int j = 0; j < m; ++j) {
            const real kj = kr[nl * m + j];
            r += R0[j * ld + i] * kj;       // symmetric matrices: column walk = row walk
            s += R1[j * ld + i] * kj;
        }
        r *= ki; s *= ki; pm = tv[i] * ki; mv = uv[i] * ki;
    }
    sc[threadIdx.x] = r; sc[SVGP_BLOCK + threadIdx.x] = s; sc[2 * SVGP_BLOCK + threadIdx.x] = pm;
    sc[3 * SVGP_BLOCK + threadIdx.x] = mv;
    __syncthreads();
    real l3 = 0, ce = 0;
    if (act && i == 0) {
        real rs = 0, ss = 0, pms = 0, mvs = 0;
#pragma unroll 8
        for (int k = 0; k < m; ++k) {
            rs += sc[nl * m + k]; ss += sc[SVGP_BLOCK + nl * m + k];
            pms += sc[2 * SVGP_BLOCK + nl * m + k]; mvs += sc[3 * SVGP_BLOCK + nl * m + k];
        }
        const size_t e = (size_t)n * a.L + l;
        const real y = a.y[e], s2 = a.s2[e], p = recip_no_nan(s2);
        const real kq = a.knn[n] - a.q[n];
        const real p_m = a.c * pms, ee = y - mvs, dd = kq + ss + ee * ee;
        real p_v = kq + rs;
        if (a.clip_pv == 1) p_v = fmin(fmax(p_v, 1e-4), 100.0);
        real ep;
        if (a.use_rng) ep = philox_normal((unsigned long long)a.state[SVGP_ST_RNG_CTR], (unsigned long long)e);
        else ep = a.eps_in[e];
        a.eps[e] = ep;
        a.p_m[e] = p_m; a.p_v[e] = p_v; a.e[e] = ee; a.d[e] = dd;
        a.z[e] = p_m + ep * sqrt(a.clip_pv == 2 ? fmin(fmax(p_v, 1e-4), 1000.0) : p_v);
        const real ls2 = log(s2);
        l3 = real(-0.5) * (p * dd + ls2);
        const real dm = p_m - y;
        ce = real(-0.5) * (real(SVGP_LOG_2PI) + ls2 + (p_v + dm * dm) * p);
    }
    l3 = block_sum(l3, red);
    ce = block_sum(ce, red);
    if (threadIdx.x == 0) {
        const size_t pi = ((size_t)l * a.nb + bx) * 2;
        a.part[pi] = l3; a.part[pi + 1] = ce;
    }
}

// =============================================================================================
// m x m factor stage, reverse (per channel) + final K_bar assembly.
// =============================================================================================
struct FactBwdArgs {
    int b_global, m, L, geco, kl_form, P;
    real c, N_train;
    const real* state;
    const real* K; const real* Ki; const real* S; const real* v; const real* Si; const real* t; const real* G;
    const real* A; const real* Aji; const real* mu; const real* u; const real* M2;
    const real* A2; const real* ud; const real* td;
    real* Kbar_part; real* Kibar_part;     // (L,m,m) each
    real* vbar; real* Ssym; real* Qm;
    real* Kbar;                             // final (m,m)
    // training step (m <= 64): workgroups [L, L + n_riders) compute the decoder's weight-gradient partials (vae_dev.hpp): work that
    // only the closing gradient reduction consumes, in the 240 CUs this launch leaves idle.  Measured (tools/decoder_split_probe.py,
    // config 2): 3 riders per image in two rounds of 2 workgroups per CU (this kernel's 200 VGPRs) 20.4 us against 18.1 without
    // riders; capped at 164 VGPRs for 3 per CU (one round) the channel workgroups themselves slow down beside two riders: 21.2 us;
    // s_setprio for the channel workgroups: no change; 1 or 2 fatter riders per image: 93 / 74 KB of LDS, a third round: 24 us.
    int n_riders;
    svgp_vae::DecWgradArgs wg;
    // single-GPU training step (round 6): the reverse statistics (A2, ud, td: P row partials per channel) ride in FRONT of the channel
    // workgroups, which wait for their own P producers on ws.flags[8 + l] (write-through payload + drained counter, as in
    // k_encoder_bwd_km); the riders wait for nobody
    int n_stat, wait_n;
    StatArgs st;
    unsigned long long* flags;
};

// STAT: 0 = the reverse statistics were a launch of their own; 1 = they ride at the head of this launch and the channel workgroup
// waits for them where Ki is last in LDS (before T1 A); 2 = ... a FIFTH LDS matrix keeps Ki for the whole workgroup (m * m <= 4 * 256
// elements, no kl_form), the wait moves behind Kbar_l = Abar G^T and the hand-off loads fly during the next two products
template <int MC, int STAT = 0>
__global__ __launch_bounds__(SVGP_BLOCK) void k_gp_factor_bwd(FactBwdArgs a) {
    extern __shared__ __align__(16) real smem[];
    int bid = (int)blockIdx.x;
    if (STAT) {
        if (bid < a.n_stat) {                  // block l P + part: the P producers of a channel are neighbours, low channels first
            const int ls = bid / a.P, part = bid - ls * a.P;
            gp_stats_block<MC, true>(a.st, part, a.P, ls, smem);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(&a.flags[8 + ls], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        bid -= a.n_stat;
    }
    if (bid >= a.L) {
        svgp_vae::decoder_wgrad_rider<SVGP_BLOCK>(a.wg, bid - a.L, smem);
        return;
    }
    const int m = MC ? MC : a.m, mp = pad16(m), ld = mp + 2, mm = mp * ld, l = bid;
    real* R0 = smem;
    real* R1 = R0 + mm;
    real* R2 = R1 + mm;
    real* R3 = R2 + mm;
    real* RK = STAT == 2 ? R3 + mm : R0;  // Ki
    // m * m <= 4 * 256 (`keep`): one more LDS matrix accumulates Kbar_l (five updates that were read-modify-writes of GLOBAL memory in
    // two thread layouts: a round trip each on the chain); it is stored once, with the last product
    const bool keep = m * m <= 4 * (int)blockDim.x;
    const int nbuf = 4 + (STAT == 2 ? 1 : 0) + (keep ? 1 : 0);
    real* RKb = smem + (nbuf - 1) * mm;
    for (int o = threadIdx.x; o < nbuf * mm; o += blockDim.x) smem[o] = 0;   // zero pads (MFMA tiles read them)
    __syncthreads();
    real* ubar = smem + nbuf * mm;      // m
    real* mubar = ubar + m;    // m
    real* tbar = mubar + m;    // m
    real* tv = tbar + m;       // m  (t_l)
    real* vv = tv + m;         // m  (v_l)
    real* muv = vv + m;        // m  (mu_hat_l)
    const size_t om = (size_t)l * m * m, ov = (size_t)l * m;
    const real gT = grad_KL_term(a.geco, a.L, a.state);
    const real g3 = svgp_seed_3(a.geco, gT), gK = svgp_seed_K(a.geco, gT, (real)a.b_global / a.N_train);
    real* Kb = a.Kbar_part + om;
    real* Kib = a.Kibar_part + om;

    const size_t sM = (size_t)a.L * m * m, sV = (size_t)a.L * m;     // strides between row partials
    mat_load(RK, ld, a.Ki, m);
    // S and A2 are needed twice each; for m <= 32 (<= 4 elements per thread) their partial sums stay in registers
    real kS[4] = {0, 0, 0, 0}, kA2[4] = {0, 0, 0, 0};
    // keep: every forward matrix the chain loads later (Aji, A, K, G, Si, M2) is fetched into registers NOW (24 reals per thread)
    Mat4 pAji = {}, pA = {}, pK = {}, pG = {}, pSi = {}, pM2 = {}, kib = {};
    real u_r = 0;
    if (keep) {
        pAji = mat_fetch4(a.Aji + om, m); pA = mat_fetch4(a.A + om, m); pK = mat_fetch4(a.K, m); pG = mat_fetch4(a.G + om, m);
        pSi = mat_fetch4(a.Si + om, m); pM2 = mat_fetch4(a.M2 + om, m);
        if (threadIdx.x < m) u_r = a.u[ov + threadIdx.x];
    }
    if (keep) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = threadIdx.x + k * blockDim.x;
            if (o < m * m) {
                kS[k] = part_sum(a.S + om + o, sM, a.P);
                if (!STAT) kA2[k] = part_sum(a.A2 + om + o, sM, a.P);
                R1[(o / m) * ld + (o % m)] = kS[k];
            }
        }
    } else {
        for (int o = threadIdx.x; o < m * m; o += blockDim.x) R1[(o / m) * ld + (o % m)] = part_sum(a.S + om + o, sM, a.P);
    }
    if (threadIdx.x < m) {
        muv[threadIdx.x] = a.mu[ov + threadIdx.x];
        if (!STAT) ubar[threadIdx.x] = part_sum(a.ud + ov + threadIdx.x, sV, a.P) +
                                       (a.kl_form ? real(0) : real(0.5) * gK * muv[threadIdx.x]);
        tv[threadIdx.x] = a.t[ov + threadIdx.x];
        vv[threadIdx.x] = part_sum(a.v + ov + threadIdx.x, sV, a.P);
    }
    __syncthreads();
    if (!STAT) mat_vec(mubar, RK, ld, ubar, m, real(1));            // Ki ubar
    mat_gemm<false, false>(R2, R1, RK, ld, m, real(1));             // T1 = S Ki
    __syncthreads();
    if (!STAT && threadIdx.x < m && !a.kl_form) mubar[threadIdx.x] += real(0.5) * gK * (keep ? u_r : a.u[ov + threadIdx.x]);
    mat_gemm<false, false>(R3, RK, R2, ld, m, real(1));             // Ki S Ki
    __syncthreads();
    if (keep) mat_put4(R1, ld, pAji, m); else mat_load(R1, ld, a.Aji + om, m);
    __syncthreads();
    for (int o = threadIdx.x; o < m * m; o += blockDim.x) {         // Abar
        const int idx = (o / m) * ld + (o % m);
        R3[idx] = real(-0.5) * g3 * R3[idx] + real(0.5) * gK * (RK[idx] - R1[idx]);
    }
    __syncthreads();
    if (keep) mat_put4(R1, ld, pA, m); else mat_load(R1, ld, a.A + om, m);
    if (STAT == 1) {
        // Everything above is a function of forward quantities alone: the reverse statistics of this channel (A2, ud, td) are awaited
        // HERE, the last point at which Ki is still in LDS for Ki ubar.  (Waiting at the top of the workgroup: 28.0 us for the launch
        // against 21.7 without the wait -- the statistics need ~6 us from launch start.)
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(&a.flags[8 + l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)a.wait_n) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 23)) { __hip_atomic_store(&a.flags[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            __hip_atomic_store(&a.flags[8 + l], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // its only consumer re-arms it
        }
        __syncthreads();
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (keep) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int o = threadIdx.x + k * blockDim.x;
                if (o < m * m) kA2[k] = part_sum<true>(a.A2 + om + o, sM, a.P);
            }
        }
        if (threadIdx.x < m)
            ubar[threadIdx.x] = part_sum<true>(a.ud + ov + threadIdx.x, sV, a.P) +
                                (a.kl_form ? real(0) : real(0.5) * gK * muv[threadIdx.x]);
        __syncthreads();
        mat_vec(mubar, R0, ld, ubar, m, real(1));                   // Ki ubar
        __syncthreads();
        if (threadIdx.x < m && !a.kl_form) mubar[threadIdx.x] += real(0.5) * gK * (keep ? u_r : a.u[ov + threadIdx.x]);
    }
    __syncthreads();
    mat_gemm<false, false>(R0, R2, R1, ld, m, real(1));             // T1 A = S Ki A   (Ki dropped)
    __syncthreads();
    real pre[4] = {0, 0, 0, 0};
    if (STAT == 2) {                                                // Kibar_l: what needs no reverse statistic (finished below)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = threadIdx.x + k * blockDim.x;
            if (o < m * m) {
                const int idx = (o / m) * ld + (o % m);
                pre[k] = -g3 * R0[idx] + real(0.5) * gK * R1[idx];
            }
        }
    } else
    for (int o = threadIdx.x; o < m * m; o += blockDim.x) {         // Kibar_l
        const int i = o / m, j = o % m, idx = i * ld + j;
        real sS, sA2;
        if (keep) {
            const int k = o / (int)blockDim.x;
            sS = k == 0 ? kS[0] : k == 1 ? kS[1] : k == 2 ? kS[2] : kS[3];
            sA2 = k == 0 ? kA2[0] : k == 1 ? kA2[1] : k == 2 ? kA2[2] : kA2[3];
        } else {
            sS = part_sum(a.S + om + o, sM, a.P); sA2 = part_sum<STAT != 0>(a.A2 + om + o, sM, a.P);
        }
        const real kv_ = -g3 * R0[idx] + real(0.5) * gK * R1[idx] + ubar[i] * muv[j] + real(0.5) * g3 * sS - sA2;
        Kib[o] = kv_;
        if (keep) {
            const int k = o / (int)blockDim.x;
            if (k == 0) kib.v[0] = kv_; else if (k == 1) kib.v[1] = kv_; else if (k == 2) kib.v[2] = kv_; else kib.v[3] = kv_;
        }
    }
    __syncthreads();
    if (a.kl_form) {
        // moving-ball KL: the summand (L/2) tr(Ki A A) feeds Abar += (gK L / 2)(Ki A + A Ki), Kibar += (gK L / 2) A A.
        // Here R1 = A, R3 = Abar; R0, R2 are free.
        const real w = real(0.5) * gK * (real)a.L;
        mat_load(R0, ld, a.Ki, m);
        __syncthreads();
        mat_gemm<false, false>(R2, R0, R1, ld, m, real(1));         // Ki A
        __syncthreads();
        for (int o = threadIdx.x; o < m * m; o += blockDim.x) {
            const int i = o / m, j = o % m;
            R3[i * ld + j] += w * (R2[i * ld + j] + R2[j * ld + i]);
        }
        mat_gemm<false, false>(R0, R1, R1, ld, m, real(1));         // A A   (Ki no longer needed)
        __syncthreads();
        for (int o = threadIdx.x; o < m * m; o += blockDim.x) Kib[o] += w * R0[(o / m) * ld + (o % m)];
        __syncthreads();
    }
    if (keep) mat_put4(R0, ld, pK, m); else mat_load(R0, ld, a.K, m);
    __syncthreads();
    mat_gemm<false, false>(R1, R0, R3, ld, m, real(1));             // Gbar = K Abar
    if (keep) mat_put4(R2, ld, pG, m); else mat_load(R2, ld, a.G + om, m);
    __syncthreads();
    if (keep) mat_gemm<false, true>(RKb, R3, R2, ld, m, real(1));    // Kbar_l = Abar G^T
    else mat_gemm_g<false, true>(Kb, R3, R2, ld, m, real(1), real(0));
    if (STAT == 2) {
        // the wait: ~9 us into the workgroup, the statistics need ~6 us from launch start
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(&a.flags[8 + l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)a.wait_n) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 23)) { __hip_atomic_store(&a.flags[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            __hip_atomic_store(&a.flags[8 + l], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // its only consumer re-arms it
        }
        __syncthreads();                                            // (also: every wave is past its reads of R2 = G)
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = threadIdx.x + k * blockDim.x;
            if (o < m * m) kA2[k] = part_sum<true>(a.A2 + om + o, sM, a.P);
        }
        real ud_r = 0, td_r = 0;
        if (threadIdx.x < m) {
            ud_r = part_sum<true>(a.ud + ov + threadIdx.x, sV, a.P);
            td_r = part_sum<true>(a.td + ov + threadIdx.x, sV, a.P);
        }
        mat_put4(R2, ld, pSi, m);
        __syncthreads();
        mat_gemm_acc<false, false>(RKb, R2, R1, ld, m, real(1));        // += Si Gbar   (STAT == 2 implies keep)
        mat_gemm<false, false>(R3, R1, R0, ld, m, real(1));             // Gbar K
        if (threadIdx.x < m) ubar[threadIdx.x] = ud_r + (a.kl_form ? real(0) : real(0.5) * gK * muv[threadIdx.x]);
        __syncthreads();
        mat_vec(mubar, RK, ld, ubar, m, real(1));                   // Ki ubar
        __syncthreads();
        if (threadIdx.x < m && !a.kl_form) mubar[threadIdx.x] += real(0.5) * gK * u_r;
        __syncthreads();
        if (threadIdx.x < m) {                                      // tbar = td + c K mubar
            real acc = 0;
            for (int j = 0; j < m; ++j) acc += R0[threadIdx.x * ld + j] * mubar[j];
            tbar[threadIdx.x] = td_r + a.c * acc;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {                               // Kibar_l, finished
            const int o = threadIdx.x + k * blockDim.x;
            if (o < m * m) { kib.v[k] = pre[k] + ubar[o / m] * muv[o % m] + real(0.5) * g3 * kS[k] - kA2[k]; Kib[o] = kib.v[k]; }
        }
        __syncthreads();
    } else {
    // tbar = td + c K mubar
    if (threadIdx.x < m) {
        real acc = 0;
        for (int j = 0; j < m; ++j) acc += R0[threadIdx.x * ld + j] * mubar[j];
        tbar[threadIdx.x] = part_sum<STAT != 0>(a.td + ov + threadIdx.x, sV, a.P) + a.c * acc;
    }
    __syncthreads();
    if (keep) mat_put4(R2, ld, pSi, m); else mat_load(R2, ld, a.Si + om, m);
    __syncthreads();
    if (keep) mat_gemm_acc<false, false>(RKb, R2, R1, ld, m, real(1));  // += Si Gbar
    else mat_gemm_g<false, false>(Kb, R2, R1, ld, m, real(1), real(1));
    mat_gemm<false, false>(R3, R1, R0, ld, m, real(1));             // Gbar K
    __syncthreads();
    }
    for (int o = threadIdx.x; o < m * m; o += blockDim.x) {
        const int i = o / m, j = o % m;
        real sA2;
        if (keep) {
            const int k = o / (int)blockDim.x;
            sA2 = k == 0 ? kA2[0] : k == 1 ? kA2[1] : k == 2 ? kA2[2] : kA2[3];
        } else {
            sA2 = part_sum<STAT != 0>(a.A2 + om + o, sM, a.P);
        }
        R3[i * ld + j] += sA2 + tbar[i] * vv[j];                    // Sibar
        if (keep) RKb[i * ld + j] += a.c * mubar[i] * tv[j];
        else Kb[o] += a.c * mubar[i] * tv[j];
    }
    if (threadIdx.x < m) {                                          // vbar = Si tbar
        real acc = 0;
        for (int j = 0; j < m; ++j) acc += R2[threadIdx.x * ld + j] * tbar[j];
        a.vbar[ov + threadIdx.x] = acc;
    }
    __syncthreads();
    mat_gemm<false, false>(R1, R2, R3, ld, m, real(1));             // Si Sibar
    __syncthreads();
    mat_gemm<false, false>(R0, R1, R2, ld, m, real(-1));            // Sgbar = -Si Sibar Si
    __syncthreads();
    for (int o = threadIdx.x; o < m * m; o += blockDim.x) {
        const int i = o / m, j = o % m;
        if (keep) RKb[i * ld + j] += R0[i * ld + j];
        else Kb[o] += R0[i * ld + j];
        const real ss = a.c * (R0[i * ld + j] + R0[j * ld + i]);
        a.Ssym[om + o] = ss;
        real m2;
        if (keep) { const int k = o / (int)blockDim.x; m2 = k == 0 ? pM2.v[0] : k == 1 ? pM2.v[1] : k == 2 ? pM2.v[2] : pM2.v[3]; }
        else m2 = a.M2[om + o];
        a.Qm[om + o] = ss - g3 * m2;
    }
    // Kbar_l -= Ki Kibar_l Ki  (the inverse's VJP is linear, so it is applied per channel here and
    // the final kernel only sums channels).  Kib was written by this workgroup above.
    __syncthreads();
    real* RKi = STAT == 2 ? RK : R1;
    if (STAT != 2) mat_load(R1, ld, a.Ki, m);
    if (keep && !a.kl_form) mat_put4(R2, ld, kib, m); else mat_load(R2, ld, Kib, m);
    __syncthreads();
    mat_gemm<false, false>(R3, RKi, R2, ld, m, real(1));
    __syncthreads();
    if (keep) mat_gemm_g_from<false, false>(Kb, RKb, R3, RKi, ld, m, real(-1));
    else mat_gemm_g<false, false>(Kb, R3, RKi, ld, m, real(-1), real(1));
}

// Kbar = sum_l Kbar_l + (L gK / 2) Ki ; one thread per element, channel loads batched.
__global__ __launch_bounds__(SVGP_BLOCK) void k_gp_factor_bwd_final(FactBwdArgs a) {
    const int m = a.m, o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= m * m) return;
    const real gT = grad_KL_term(a.geco, a.L, a.state);
    const real gK = svgp_seed_K(a.geco, gT, (real)a.b_global / a.N_train);
    real s = 0;
#pragma unroll 8
    for (int l = 0; l < a.L; ++l) s += a.Kbar_part[(size_t)l * m * m + o];
    a.Kbar[o] = s + real(0.5) * gK * (real)a.L * a.Ki[o];
}

// =============================================================================================
// per-sample stage, reverse.  Pass 1 grid (ceil(b/RB), L): per-channel row terms.
// Pass 2 grid ceil(b/RB): sum over channels + the Ki term.
// =============================================================================================
struct PostBwdArgs {
    int b, m, L, geco;
    real c;
    const real* state;
    const real* Kn; const real* y; const real* s2; const real* p_m; const real* p_v; const real* e; const real* d;
    const real* g_pv; const real* g_pm; const real* mvbar;
    const real* Si; const real* Qm; const real* Ssym; const real* u; const real* t; const real* vbar; const real* Ki;
    real* Knbar_part;   // (L, b, m)
    real* Knbar; real* knnbar; real* ybar; real* s2bar;
    // training phases: workgroups [nb_rows, nb_rows + n_final) of k_gp_posterior_bwd_sum do k_gp_factor_bwd_final's work
    // (Kbar = sum_l Kbar_l + (L gK / 2) Ki; independent of this stage, consumed by the kernel-matrix reverse pass)
    int nb_rows, n_final, b_global;
    real N_train;
    const real* Kbar_part; real* Kbar;
};

template <int MC>
__global__ __launch_bounds__(SVGP_BLOCK) void k_gp_posterior_bwd_l(PostBwdArgs a) {
    extern __shared__ __align__(16) real smem[];
    const int m = MC ? MC : a.m, ld = m + 1, mm = m * ld, l = blockIdx.y;
    real* R0 = smem;            // Si_l
    real* R1 = R0 + mm;         // Q_l
    real* R2 = R1 + mm;         // Ssym_l
    real* uv = R2 + mm;         // u_l
    real* tv = uv + m;          // t_l
    real* vb = tv + m;          // vbar_l
    real* kr = vb + m;          // RB x m
    real* sc = kr + SVGP_BLOCK; // 2 x 256
    const int RB = blockDim.x / m;
    const int nl = threadIdx.x / m, i = threadIdx.x % m, n = blockIdx.x * RB + nl;
    const bool act = nl < RB && n < a.b;
    const size_t om = (size_t)l * m * m, ov = (size_t)l * m;
    mat_load(R0, ld, a.Si + om, m);
    mat_load(R1, ld, a.Qm + om, m);
    mat_load(R2, ld, a.Ssym + om, m);
    if (threadIdx.x < m) { uv[threadIdx.x] = a.u[ov + threadIdx.x]; tv[threadIdx.x] = a.t[ov + threadIdx.x]; vb[threadIdx.x] = a.vbar[ov + threadIdx.x]; }
    if (act) kr[nl * m + i] = a.Kn[(size_t)n * m + i];
    __syncthreads();
    const real gT = grad_KL_term(a.geco, a.L, a.state);
    const real g3 = svgp_seed_3(a.geco, gT);
    real ksk = 0, kv = 0;
    if (act) {
        const size_t e = (size_t)n * a.L + l;
        const real p = recip_no_nan(a.s2[e]), gpv = a.g_pv[e];
        real sik = 0, qk = 0, ssk = 0;
#pragma unroll 8
        for (int j = 0; j < m; ++j) {
            const real kj = kr[nl * m + j];
            sik += R0[j * ld + i] * kj;
            qk += R1[j * ld + i] * kj;
            ssk += R2[j * ld + i] * kj;
        }
        const real ki = kr[nl * m + i];
        a.Knbar_part[((size_t)l * a.b + n) * m + i] = real(2) * gpv * sik + p * qk + a.mvbar[e] * uv[i] +
                                                      a.c * a.g_pm[e] * tv[i] + p * a.y[e] * vb[i];
        ksk = real(0.5) * ssk * ki;
        kv = vb[i] * ki;
    }
    sc[threadIdx.x] = ksk; sc[SVGP_BLOCK + threadIdx.x] = kv;
    __syncthreads();
    if (act && i == 0) {
        real kS = 0, kV = 0;
        for (int k = 0; k < m; ++k) { kS += sc[nl * m + k]; kV += sc[SVGP_BLOCK + nl * m + k]; }
        const size_t e = (size_t)n * a.L + l;
        const real y = a.y[e], s2 = a.s2[e], p = recip_no_nan(s2);
        const real dm = a.p_m[e] - y;
        const real pbar = real(-0.5) * g3 * a.d[e] + kS + y * kV;
        const real ce_y = -gT * p * dm;
        const real ce_s2 = real(0.5) * gT * (p - (a.p_v[e] + dm * dm) * p * p);
        a.ybar[e] = ce_y - g3 * p * a.e[e] + p * kV;
        a.s2bar[e] = ce_s2 - real(0.5) * g3 * p - pbar * p * p;
    }
}

// Pass 2 as a device function: block `bid` of nb_rows + n_final, `nthr` live threads (the stand-alone launch: blockDim.x; as rider
// workgroups of svgp_mnist_encoder_bwd_km: the first SVGP_BLOCK threads of a VAE_NT-thread workgroup)
template <int MC, bool COH = false>
__device__ __forceinline__ void posterior_bwd_sum_block(const PostBwdArgs& a, int bid, int nthr, real* smem) {
    if (a.n_final > 0 && bid >= a.nb_rows) {
        const int m = MC ? MC : a.m, o = (bid - a.nb_rows) * nthr + threadIdx.x;
        if (o >= m * m) return;
        const real gT = grad_KL_term(a.geco, a.L, a.state);
        const real gK = svgp_seed_K(a.geco, gT, (real)a.b_global / a.N_train);
        real s = 0;
#pragma unroll 8
        for (int l = 0; l < a.L; ++l) s += a.Kbar_part[(size_t)l * m * m + o];
        st_co<COH>(a.Kbar + o, s + real(0.5) * gK * (real)a.L * a.Ki[o]);
        return;
    }
    const int m = MC ? MC : a.m, ld = m + 1, mm = m * ld;
    real* R0 = smem;            // Ki
    real* kr = R0 + mm;         // RB x m
    real* qb = kr + SVGP_BLOCK; // RB
    const int RB = nthr / m;
    const int nl = threadIdx.x / m, i = threadIdx.x % m, n = bid * RB + nl;
    const bool act = nl < RB && n < a.b;
    mat_load_nt(R0, ld, a.Ki, m, nthr);
    if (act) kr[nl * m + i] = a.Kn[(size_t)n * m + i];
    const real gT = grad_KL_term(a.geco, a.L, a.state);
    // qbar_n = sum_l [g3 / (2 s2_nl) - g_pv_nl]: the L terms of a row are loaded by L of its m threads at once (one thread walking
    // them was a chain of L dependent L2 round trips: ~6 us of this 3 us block), then added in channel order by the row's first thread
    real* qt = qb + RB;         // RB x L terms
    if (act)
        for (int l = i; l < a.L; l += m) {
            const size_t e = (size_t)n * a.L + l;
            qt[nl * a.L + l] = real(0.5) * svgp_seed_3(a.geco, gT) * recip_no_nan(a.s2[e]) - a.g_pv[e];
        }
    __syncthreads();
    if (act && i == 0) {
        real qbar = 0;
        for (int l = 0; l < a.L; ++l) qbar += qt[nl * a.L + l];
        qb[nl] = qbar;
        st_co<COH>(a.knnbar + n, -qbar);
    }
    __syncthreads();
    if (act) {
        real acc = 0;
#pragma unroll 8
        for (int l = 0; l < a.L; ++l) acc += a.Knbar_part[((size_t)l * a.b + n) * m + i];
        real w = 0;
#pragma unroll 8
        for (int j = 0; j < m; ++j) w += R0[j * ld + i] * kr[nl * m + j];
        st_co<COH>(a.Knbar + (size_t)n * m + i, acc + real(2) * qb[nl] * w);
    }
}

template <int MC>
__global__ __launch_bounds__(SVGP_BLOCK) void k_gp_posterior_bwd_sum(PostBwdArgs a) {
    extern __shared__ __align__(16) real smem[];
    posterior_bwd_sum_block<MC>(a, (int)blockIdx.x, (int)blockDim.x, smem);
}

// Training step (m <= 64, round 6): the kernel-matrix VJP and the encoder's reverse pass are independent (both consume the
// reverse row stage), so they share ONE launch of VAE_NT-thread workgroups: the m + nrb VJP workgroups come FIRST in the grid and
// run on their first SVGP_BLOCK threads (the other waves exit at once), the image workgroups follow.  168 VGPRs and 75 KB of LDS
// let a VJP workgroup and an image workgroup share a CU, so no image workgroup waits for a VJP workgroup to retire.
// Round 6, second step: pass 2 of the reverse row stage (sum over channels: Knbar, knnbar, Kbar -- consumed by the VJP only) rides
// in the same launch, in FRONT of the VJP workgroups, which wait for it on a counter in the workspace: the n_sum producers write
// their results THROUGH to memory (sc1 stores), drain them (s_waitcnt) and increment the counter; the VJP workgroups poll it and
// read the payload past their XCD's L2 (sc1 loads, ld_co); the last of them through the gate resets both counters.  Workgroup ids are dealt to the XCDs round-robin and dispatched in order per XCD, and all n_sum + n_km + n_img
// workgroups fit the chip at once (two per CU), so a waiting VJP workgroup never holds a slot a producer needs.  The image
// workgroups (the longest of the launch) do not wait: the sum -> VJP chain (~10 us) hides under them.
// (No occupancy hint in the launch bounds: the cfg-2 instance allocates 161 VGPRs <= 168 on its own, and `__launch_bounds__(VAE_NT, 3)`
// made the scheduler trade the image code's instruction-level parallelism for registers it did not need: 17.8 against 16.1 us for the
// 256 image workgroups alone, 21.6 against 19.9 us for this launch.)
template <int MC, int MM, bool SUM>
__global__ __launch_bounds__(VAE_NT) void k_encoder_bwd_km(svgp_vae::EncBwdArgs e, int n_km, KernArgs a, real rep_weight,
                                                                 int train_ip, const real* __restrict__ K,
                                                                 const real* __restrict__ Kn, const real* Kbar,
                                                                 const real* Knbar, const real* knnbar,
                                                                 const real* __restrict__ knn,
                                                                 real* __restrict__ d_ip, real* __restrict__ d_on,
                                                                 real* __restrict__ part_gp, int n_sum, PostBwdArgs pb,
                                                                 unsigned long long* __restrict__ flags) {
    extern __shared__ __align__(16) real smem[];
    int bid = (int)blockIdx.x;
    {
        const int n_front = n_km + (SUM ? n_sum : 0);
        if (bid >= n_front) {
            // (image waves before the VJP / sum waves sharing their SIMDs: 20.1 -> 19.8 us; the other way round: no gain)
            __builtin_amdgcn_s_setprio(3);
            svgp_vae::encoder_bwd_images<VAE_NT>(e, bid - n_front, (int)gridDim.x - n_front, smem);
            return;
        }
    }
    if (SUM) {
        if (bid < n_sum) {
            if (threadIdx.x >= SVGP_BLOCK) return;
            posterior_bwd_sum_block<MC, true>(pb, bid, SVGP_BLOCK, smem);      // Knbar, knnbar, Kbar: written through (sc1)
            __builtin_amdgcn_s_waitcnt(0);                                    // this wave's stores have left
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(&flags[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        bid -= n_sum;
    }
    if (bid < n_km) {
        if (threadIdx.x >= SVGP_BLOCK) return;
        if (SUM) {
            if (threadIdx.x == 0) {
                // (bounded: ~1 s of polling, then the sticky error word flags[2] is set and the workgroup goes on -- a lost producer
                // must show up as a wrong, flagged result, never as a hung GPU)
                int spins = 0;
                while (__hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)n_sum) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1 << 22)) { __hip_atomic_store(&flags[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                // the last VJP workgroup through the gate re-arms it for the next step
                if (__hip_atomic_fetch_add(&flags[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)(n_km - 1)) {
                    __hip_atomic_store(&flags[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&flags[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();
            __atomic_signal_fence(__ATOMIC_SEQ_CST);                          // (compiler: no payload load above the gate)
        }
        const int m = MC ? MC : a.m;
        if (bid < m) km_bwd_cols<MC, MM, SVGP_BLOCK, SUM>(bid, a, rep_weight, train_ip, K, Kn, Kbar, Knbar, d_ip, part_gp);
        else km_bwd_rows<MC, MM, SVGP_BLOCK, SUM>(bid - m, a, Knbar, knnbar, knn, d_on, part_gp);
        return;
    }
}

// Training step, m <= 32 (round 6): the deferred tail of the forward factor stage -- Aji = (A_hat + jitter I)^-1 and the log det term of
// KL (SVGPVAE_model.py:271-279), needed by the reverse factor stage and the KL scalar only -- as L rider workgroups at the HEAD of the
// decoder's data-reverse launch (first SVGP_BLOCK threads; the single-wave sweep of sweep32.hpp).  In the forward row-stage launch,
// where it rode before, the 4.4 us sweep + its loads bounded the launch: 10.1 us against 7.8 without it.  74 KB of LDS and <= 128
// VGPRs let a rider and an image workgroup share a CU, so the 256 image workgroups still start at once.
struct AjiArgs {
    int m, L;
    real jitter;
    const real* Ahat; real* Aji; real* KL;
};
__global__ __launch_bounds__(VAE_NT) void k_decoder_bwd_data_aji(svgp_vae::DecBwdDataArgs d, AjiArgs a) {
    extern __shared__ __align__(16) real smem[];
    const int l = blockIdx.x;
    if (l < a.L) {
        if (threadIdx.x >= SVGP_BLOCK) return;
        const int m = a.m, ld = m + 1;
        real* A = smem;
        real* W = A + m * ld;
        mat_load_nt(A, ld, a.Ahat + (size_t)l * m * m, m, SVGP_BLOCK);
        __syncthreads();
        if ((int)threadIdx.x < m) A[threadIdx.x * ld + threadIdx.x] += a.jitter;
        const real ldA = chol_inv(A, W, ld, m);                 // m <= 32: wave 0 alone, any number of live waves
        for (int o = threadIdx.x; o < m * m; o += SVGP_BLOCK) a.Aji[(size_t)l * m * m + o] = A[(o / m) * ld + (o % m)];
        if (threadIdx.x == 0) a.KL[l] -= real(0.5) * ldA;
        return;
    }
    // the image waves before a rider's sweep wave on the same SIMD: without it the 16 image workgroups that share a CU with a rider
    // finish 2 us late (16.8 against 14.7 us for the launch; 14.5 without riders)
    __builtin_amdgcn_s_setprio(3);
    svgp_vae::decoder_bwd_data_images<true>(d, l - a.L, (int)gridDim.x - a.L, smem);
}

template <typename F>
int set_dyn_lds(F kernel, size_t bytes) {
    SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return SVGP_OK;
}
// launch of a kernel template <int MC> (compile-time m): the m = 32 instance for config 2, the run-time-m instance otherwise
#define LAUNCH_MC(kern, m_, grid_, lds_, stream_, args_)                                                              \
    do {                                                                                                              \
        int rc_mc = (m_) == 32 ? set_dyn_lds(kern<32>, lds_) : set_dyn_lds(kern<0>, lds_);                             \
        if (rc_mc) return rc_mc;                                                                                      \
        if ((m_) == 32) hipLaunchKernelGGL(kern<32>, grid_, dim3(SVGP_BLOCK), lds_, (hipStream_t)(stream_), args_);    \
        else hipLaunchKernelGGL(kern<0>, grid_, dim3(SVGP_BLOCK), lds_, (hipStream_t)(stream_), args_);                \
        SVGP_LAUNCH_CHECK();                                                                                          \
    } while (0)

KernArgs make_kern_args(const svgp_mnist_cfg* c, const svgp_mnist_param_layout& pl, const double* theta,
                        const double* aux) {
    return svgp_make_kern_args(c, pl, theta, aux);
}

}  // namespace

#define GET_LAYOUTS()                                          \
    svgp_mnist_param_layout pl;                                \
    svgp_mnist_ws_layout wl;                                   \
    {                                                          \
        int rc_ = svgp_mnist_param_layout_get(c, &pl);         \
        if (rc_) return rc_;                                   \
        rc_ = svgp_mnist_ws_layout_get(c, &wl);                \
        if (rc_) return rc_;                                   \
    }

static inline size_t mat_lds(int m, int nmat) { return (size_t)nmat * m * (m + 1) * sizeof(real); }
static inline size_t mat_lds_pad(int m, int nmat) {
    const int mp = (m + 15) & ~15;
    return (size_t)nmat * mp * (mp + 2) * sizeof(real);
}

extern "C" int svgp_kernel_matrix_fwd(const svgp_mnist_cfg* c, const double* theta, const double* aux, double* ws,
                                      void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(theta && aux && ws, SVGP_ERR_INVALID, "NULL device pointer");
    KernArgs a = make_kern_args(c, pl, theta, aux);
    const long long total = (long long)c->b * c->m + (long long)c->m * c->m + c->b;
    hipLaunchKernelGGL(k_kernel_matrix_fwd, dim3((unsigned)((total + SVGP_BLOCK - 1) / SVGP_BLOCK)), dim3(SVGP_BLOCK), 0,
                       (hipStream_t)stream, a, ws + wl.K, ws + wl.Kn, ws + wl.knn);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_kernel_matrix_xy(int M, int normalize, int nx, const double* x, int x_gather, int ny, const double* y,
                                     int y_gather, const double* table, const double* l_GP, const double* amplitude,
                                     int diag_only, double* out, void* stream) {
    SVGP_REQUIRE(M >= 1 && nx >= 1 && ny >= 1, SVGP_ERR_INVALID, "bad shape M=%d nx=%d ny=%d", M, nx, ny);
    SVGP_REQUIRE(x && y && l_GP && amplitude && out, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(!(x_gather || y_gather) || table, SVGP_ERR_INVALID, "a gathered side needs the object-vector table");
    SVGP_REQUIRE(!diag_only || nx == ny, SVGP_ERR_INVALID, "diag_only needs nx == ny (nx=%d ny=%d)", nx, ny);
    KernXYArgs a;
    a.M = M; a.normalize = normalize; a.nx = nx; a.ny = ny; a.xg = x_gather; a.yg = y_gather; a.diag = diag_only;
    a.x = x; a.y = y; a.table = table; a.ls = l_GP; a.amp = amplitude;
    const long long total = diag_only ? (long long)nx : (long long)nx * ny;
    hipLaunchKernelGGL(k_kernel_matrix_xy, dim3((unsigned)((total + SVGP_BLOCK - 1) / SVGP_BLOCK)), dim3(SVGP_BLOCK), 0,
                       (hipStream_t)stream, a, out);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

static int kernel_matrix_bwd_impl(const svgp_mnist_cfg* c, const double* theta, const double* aux, double* ws,
                                  bool scatter, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(theta && aux && ws, SVGP_ERR_INVALID, "NULL device pointer");
    KernArgs a = make_kern_args(c, pl, theta, aux);
    real* grad = ws + wl.grad;
    const int RBk = svgp_rows_per_block(c), nrb = (c->b + RBk - 1) / RBk;
    const size_t lds_rows = (size_t)((km_stage_O(c->m, c->M) ? c->m * c->M : c->m) + RBk * c->m + RBk * c->M) * sizeof(real);
    const bool cfg2_shape = c->m == 32 && c->M == 8, cfg3_shape = c->m == 256 && c->M == 32;
    {
        int rc_ = cfg2_shape ? set_dyn_lds(k_kernel_matrix_bwd_cr<32, 8>, lds_rows)
                  : cfg3_shape ? set_dyn_lds(k_kernel_matrix_bwd_cr<256, 32>, lds_rows)
                               : set_dyn_lds(k_kernel_matrix_bwd_cr<0, 0>, lds_rows);
        if (rc_) return rc_;
    }
    // m > 64: svgp_big_factor_bwd has applied cfg.rep_weight to the replicated part of Kbar and added the rank-local row sums
    // unweighted (gp_large.hip header), so Kbar is taken as it is
    const real km_rep_weight = c->m > SVGP_M_MAX ? real(1) : (real)c->rep_weight;
#define KM_BWD_ARGS a, km_rep_weight, c->train_ip, ws + wl.K, ws + wl.Kn, ws + wl.Kbar, ws + wl.Knbar, ws + wl.knnbar, ws + wl.knn, \
                    grad + pl.ip, ws + wl.d_on, ws + wl.part_gp
    if (cfg2_shape) hipLaunchKernelGGL((k_kernel_matrix_bwd_cr<32, 8>), dim3(c->m + nrb), dim3(SVGP_BLOCK), lds_rows, (hipStream_t)stream, KM_BWD_ARGS);
    else if (cfg3_shape) hipLaunchKernelGGL((k_kernel_matrix_bwd_cr<256, 32>), dim3(c->m + nrb), dim3(SVGP_BLOCK), lds_rows, (hipStream_t)stream, KM_BWD_ARGS);
    else hipLaunchKernelGGL((k_kernel_matrix_bwd_cr<0, 0>), dim3(c->m + nrb), dim3(SVGP_BLOCK), lds_rows, (hipStream_t)stream, KM_BWD_ARGS);
#undef KM_BWD_ARGS
    SVGP_LAUNCH_CHECK();
    if (scatter) {
        const int n_ov = c->n_obj * c->M;
        hipLaunchKernelGGL(k_kernel_matrix_bwd_scatter, dim3((n_ov + SVGP_BLOCK - 1) / SVGP_BLOCK + 1), dim3(SVGP_BLOCK),
                           svgp_km_scatter_lds(c->b, c->M), (hipStream_t)stream, a, c->m + nrb, c->train_gp, c->train_ov,
                           ws + wl.d_on, ws + wl.part_gp, grad + pl.ov, grad + pl.l_GP, grad + pl.amplitude);
        SVGP_LAUNCH_CHECK();
    }
    return SVGP_OK;
}

// svgp_kernel_matrix_bwd_partials + svgp_mnist_encoder_bwd in ONE launch (m <= 64; see k_encoder_bwd_km); with_sum: + pass 2 of the
// reverse row stage in front of them (svgp_gp_posterior_bwd_rows must have run; ws.flags must be zero on entry and is left zero)
static PostBwdArgs make_pb(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, double* ws, const double* state, bool with_final);
static int encoder_bwd_km_impl(const svgp_mnist_cfg* c, const double* theta, const double* images, const double* aux, double* ws,
                               const double* state, bool with_sum, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(theta && images && aux && ws && (state || !with_sum), SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m <= SVGP_M_MAX, SVGP_ERR_UNSUPPORTED, "the merged launch exists for m <= %d", SVGP_M_MAX);
    KernArgs a = make_kern_args(c, pl, theta, aux);
    real* grad = ws + wl.grad;
    const int RBk = svgp_rows_per_block(c), nrb = (c->b + RBk - 1) / RBk, n_km = c->m + nrb;
    const size_t lds_rows = (size_t)((km_stage_O(c->m, c->M) ? c->m * c->M : c->m) + RBk * c->m + RBk * c->M) * sizeof(real);
    size_t lds = (size_t)svgp_vae::enc_bwd_lds((int)pl.n_enc) * sizeof(real);
    if (lds_rows > lds) lds = lds_rows;
    PostBwdArgs pb;
    memset(&pb, 0, sizeof(pb));
    int n_sum = 0;
    if (with_sum) {
        pb = make_pb(c, wl, ws, state, true);
        n_sum = pb.nb_rows + pb.n_final;
        const size_t lds2 = mat_lds(c->m, 1) + (size_t)(SVGP_BLOCK + (SVGP_BLOCK / c->m) * (1 + c->L)) * sizeof(real);
        if (lds2 > lds) lds = lds2;
    }
    const bool cfg2_shape = c->m == 32 && c->M == 8;
    int rc = cfg2_shape ? (with_sum ? set_dyn_lds(k_encoder_bwd_km<32, 8, true>, lds) : set_dyn_lds(k_encoder_bwd_km<32, 8, false>, lds))
                        : (with_sum ? set_dyn_lds(k_encoder_bwd_km<0, 0, true>, lds) : set_dyn_lds(k_encoder_bwd_km<0, 0, false>, lds));
    if (rc) return rc;
    const svgp_vae::EncBwdArgs e = svgp_make_enc_bwd_args(c, wl, theta, images, ws);
    const dim3 grid(n_sum + n_km + svgp_n_part(c));
#define KM_BWD_ARGS e, n_km, a, (real)c->rep_weight, c->train_ip, ws + wl.K, ws + wl.Kn, ws + wl.Kbar, ws + wl.Knbar, ws + wl.knnbar, \
                    ws + wl.knn, grad + pl.ip, ws + wl.d_on, ws + wl.part_gp, n_sum, pb, reinterpret_cast<unsigned long long*>(ws + wl.flags)
    if (cfg2_shape && with_sum) hipLaunchKernelGGL((k_encoder_bwd_km<32, 8, true>), grid, dim3(VAE_NT), lds, (hipStream_t)stream, KM_BWD_ARGS);
    else if (cfg2_shape) hipLaunchKernelGGL((k_encoder_bwd_km<32, 8, false>), grid, dim3(VAE_NT), lds, (hipStream_t)stream, KM_BWD_ARGS);
    else if (with_sum) hipLaunchKernelGGL((k_encoder_bwd_km<0, 0, true>), grid, dim3(VAE_NT), lds, (hipStream_t)stream, KM_BWD_ARGS);
    else hipLaunchKernelGGL((k_encoder_bwd_km<0, 0, false>), grid, dim3(VAE_NT), lds, (hipStream_t)stream, KM_BWD_ARGS);
#undef KM_BWD_ARGS
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
// svgp_mnist_decoder_bwd_data_pre + the deferred (A_hat + jI)^-1 of svgp_gp_factor_fwd_defer_aji in one launch (m <= 32); the forward row
// stage then runs WITHOUT its riders (svgp_gp_posterior_fwd).  Same bits as svgp_gp_posterior_fwd_with_aji + svgp_mnist_decoder_bwd_data_pre.
extern "C" int svgp_mnist_decoder_bwd_data_pre_aji(const svgp_mnist_cfg* c, const double* theta, const double* images, double* ws,
                                                   const double* state, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(theta && images && ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m <= 32, SVGP_ERR_UNSUPPORTED, "the rider inverse is the single-wave sweep (m <= 32), m = %d", c->m);
    const svgp_vae::DecBwdDataArgs d = svgp_make_dec_bwd_data_args(c, wl, theta, images, ws, state);
    AjiArgs a;
    a.m = c->m; a.L = c->L; a.jitter = c->jitter; a.Ahat = ws + wl.A; a.Aji = ws + wl.Aji; a.KL = ws + wl.KL;
    size_t lds = (size_t)svgp_vae::dec_bwd_data_lds(c->L, (int)(pl.n_vae - pl.n_enc), true) * sizeof(real);
    const size_t lds_inv = mat_lds(c->m, 1) + (size_t)(5 * c->m + 80) * sizeof(real);
    if (lds_inv > lds) lds = lds_inv;
    int rc = set_dyn_lds(k_decoder_bwd_data_aji, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(k_decoder_bwd_data_aji, dim3(a.L + svgp_n_part(c)), dim3(VAE_NT), lds, (hipStream_t)stream, d, a);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_mnist_decoder_bwd_data_aji_regs(int* out) {
    SVGP_REQUIRE(out, SVGP_ERR_INVALID, "out is NULL");
    hipFuncAttributes fa;
    SVGP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_decoder_bwd_data_aji)));
    *out = fa.numRegs;
    return SVGP_OK;
}
// Registers per lane of the config-2 instance of the merged launch (hipFuncGetAttributes): a VJP workgroup and an image workgroup
// share a CU only while it is <= 168 (three waves per SIMD); tests assert that, since no launch-bounds hint enforces it.
extern "C" int svgp_mnist_encoder_bwd_km_regs(int* out) {
    SVGP_REQUIRE(out, SVGP_ERR_INVALID, "out is NULL");
    hipFuncAttributes fa;
    SVGP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_encoder_bwd_km<32, 8, true>)));
    *out = fa.numRegs;
    return SVGP_OK;
}
extern "C" int svgp_mnist_encoder_bwd_km(const svgp_mnist_cfg* c, const double* theta, const double* images, const double* aux,
                                         double* ws, void* stream) {
    return encoder_bwd_km_impl(c, theta, images, aux, ws, nullptr, false, stream);
}
extern "C" int svgp_mnist_encoder_bwd_km_sum(const svgp_mnist_cfg* c, const double* theta, const double* images, const double* aux,
                                             double* ws, const double* state, void* stream) {
    return encoder_bwd_km_impl(c, theta, images, aux, ws, state, true, stream);
}

extern "C" int svgp_kernel_matrix_bwd(const svgp_mnist_cfg* c, const double* theta, const double* aux, double* ws,
                                      void* stream) {
    return kernel_matrix_bwd_impl(c, theta, aux, ws, true, stream);
}

// everything except the object-table scatter and the two scalar sums, which svgp_mnist_grad_reduce_all performs
extern "C" int svgp_kernel_matrix_bwd_partials(const svgp_mnist_cfg* c, const double* theta, const double* aux,
                                               double* ws, void* stream) {
    return kernel_matrix_bwd_impl(c, theta, aux, ws, false, stream);
}

static StatArgs make_stat_args(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, double* ws, const double* state,
                               int mode, int with_aji, size_t* lds_out) {
    StatArgs a;
    memset(&a, 0, sizeof(a));
    a.b = c->b; a.m = c->m; a.L = c->L; a.mode = mode; a.c = c->N_train / (double)c->b_global; a.jitter = c->jitter;
    a.geco = SVGP_LOSS_FLAGS(c); a.clip_pv = c->clip_pv;
    a.Kn = ws + wl.Kn; a.y = ws + wl.qnet_mu; a.s2 = ws + wl.qnet_var;
    a.p_m = ws + wl.p_m; a.p_v = ws + wl.p_v; a.e = ws + wl.e; a.eps = ws + wl.eps; a.zbar = ws + wl.zbar;
    a.state = state;
    a.g_pv = ws + wl.g_pv; a.g_pm = ws + wl.g_pm; a.mvbar = ws + wl.mvbar;
    if (mode == 0) { a.S = ws + wl.S; a.v1 = ws + wl.v; a.v2 = nullptr; }
    else { a.S = ws + wl.A2; a.v1 = ws + wl.ud; a.v2 = ws + wl.td; }
    a.K = ws + wl.K; a.Ki = ws + wl.Ki; a.ldK = ws + wl.ldK;
    a.with_aji = with_aji; a.Ahat = ws + wl.A; a.Aji = ws + wl.Aji; a.KL = ws + wl.KL;
    const int m = c->m;
    const int mp_ = (m + 15) & ~15;
    a.rc_rows = (8192 / (mp_ + 2)) & ~3;        // <= 64 KB tile of K_nm rows per pass, multiple of 4
    const int P = svgp_stat_parts(c), RPh = ((c->b + P - 1) / P + 3) & ~3;
    if (a.rc_rows > RPh) a.rc_rows = RPh;
    size_t lds = (size_t)(a.rc_rows * (mp_ + 2) + 3 * a.rc_rows + 2 * SVGP_BLOCK) * sizeof(real);
    const size_t lds_inv = mat_lds(m, 1) + (size_t)(5 * m + 80) * sizeof(real);
    if ((mode == 0 || with_aji) && lds_inv > lds) lds = lds_inv;
    *lds_out = lds;
    return a;
}
static int launch_stats(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, double* ws, const double* state,
                        int mode, int with_aji, void* stream) {
    size_t lds = 0;
    StatArgs a = make_stat_args(c, wl, ws, state, mode, with_aji, &lds);
    const int m = c->m, P = svgp_stat_parts(c);
    LAUNCH_MC(k_gp_stats, m, dim3(P, c->L + (mode == 0 ? 1 : (with_aji ? c->L : 0))), lds, stream, a);
    return SVGP_OK;
}

extern "C" int svgp_gp_stats_fwd(const svgp_mnist_cfg* c, double* ws, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws, SVGP_ERR_INVALID, "NULL device pointer");
    if (c->m > SVGP_M_MAX) return svgp_big_stats(c, wl, ws, nullptr, 0, stream);
    return launch_stats(c, wl, ws, nullptr, 0, 0, stream);
}

extern "C" int svgp_gp_stats_bwd(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    if (c->m > SVGP_M_MAX) return svgp_big_stats(c, wl, ws, state, 1, stream);
    return launch_stats(c, wl, ws, state, 1, 0, stream);
}

// Training-phase pair (m <= 64): the factor stage without its last inverse, and the backward statistics launch with
// L extra workgroups that finish it.  Same results as svgp_gp_factor_fwd ... svgp_gp_stats_bwd.
extern "C" int svgp_gp_stats_bwd_with_aji(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    if (c->m > SVGP_M_MAX) return svgp_big_stats(c, wl, ws, state, 1, stream);
    return launch_stats(c, wl, ws, state, 1, 1, stream);
}

static inline int rows_per_block(int m) { return SVGP_BLOCK / m; }

static int factor_fwd_impl(const svgp_mnist_cfg* c, double* ws, int defer_aji, void* stream);
extern "C" int svgp_gp_factor_fwd(const svgp_mnist_cfg* c, double* ws, void* stream) {
    return factor_fwd_impl(c, ws, 0, stream);
}
extern "C" int svgp_gp_factor_fwd_defer_aji(const svgp_mnist_cfg* c, double* ws, void* stream) {
    return factor_fwd_impl(c, ws, 1, stream);
}
// m > 64: the tail that svgp_gp_factor_fwd_defer_aji leaves out -- (A_hat_l + jI)^-1, its log det, KL_l; may run on another
// stream than the stages that follow (it shares no buffer with the row stage, the decoder and the reverse statistics)
extern "C" int svgp_gp_factor_fwd_aji_tail(const svgp_mnist_cfg* c, double* ws, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX, SVGP_ERR_UNSUPPORTED,
                 "for m <= %d the deferred inverse rides in svgp_gp_posterior_fwd_with_aji / svgp_gp_stats_bwd_with_aji", SVGP_M_MAX);
    return svgp_big_factor_fwd(c, wl, ws, stream, 0, c->L, 2);
}
// internal (api.hip): one part of the large-m forward factor stage (gp_large.hip svgp_big_factor_fwd: 5 = the channel-independent
// block, 6 = the channel block up to mu, 7 = u and the KL terms)
int svgp_gp_factor_fwd_part(const svgp_mnist_cfg* c, double* ws, void* stream, int part) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX && c->m < SVGP_CHOL_INVERSE_MIN_M && part >= 5 && part <= 7, SVGP_ERR_UNSUPPORTED,
                 "the split forward factor stage exists for %d < m < %d", SVGP_M_MAX, SVGP_CHOL_INVERSE_MIN_M);
    return svgp_big_factor_fwd(c, wl, ws, stream, 0, c->L, part);
}
// channel windows of the factor stages (large-m path): see svgp_big_factor_fwd
extern "C" int svgp_gp_factor_fwd_channels(const svgp_mnist_cfg* c, int l0, int nl, double* ws, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX, SVGP_ERR_UNSUPPORTED,
                 "channel windows exist for the large-m path (m > %d); below it the factor stage is L small workgroups", SVGP_M_MAX);
    SVGP_REQUIRE(l0 >= 0 && nl >= 1 && l0 + nl <= c->L, SVGP_ERR_INVALID, "channel window [%d, %d) outside 0..%d", l0, l0 + nl, c->L);
    return svgp_big_factor_fwd(c, wl, ws, stream, l0, nl);
}
extern "C" int svgp_gp_factor_bwd_channels(const svgp_mnist_cfg* c, int l0, int nl, double* ws, const double* state,
                                           void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX, SVGP_ERR_UNSUPPORTED,
                 "channel windows exist for the large-m path (m > %d); below it the factor stage is L small workgroups", SVGP_M_MAX);
    SVGP_REQUIRE(l0 >= 0 && nl >= 1 && l0 + nl <= c->L, SVGP_ERR_INVALID, "channel window [%d, %d) outside 0..%d", l0, l0 + nl, c->L);
    return svgp_big_factor_bwd(c, wl, ws, state, stream, l0, nl);
}

// ... and their parts (the split of the two-stream step on a window): forward part 0 = whole stage, 1 = without the
// (A_hat + jI)^-1 tail, 2 = the tail; reverse part 0 = whole stage, 1 = early half, 2 = late half, 3 / 4 = the two parts of the
// early half (svgp_gp_factor_bwd_early_a / _b)
extern "C" int svgp_gp_factor_fwd_channels_part(const svgp_mnist_cfg* c, int l0, int nl, int part, double* ws, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX, SVGP_ERR_UNSUPPORTED, "channel windows exist for the large-m path (m > %d)", SVGP_M_MAX);
    SVGP_REQUIRE(l0 >= 0 && nl >= 1 && l0 + nl <= c->L, SVGP_ERR_INVALID, "channel window [%d, %d) outside 0..%d", l0, l0 + nl, c->L);
    SVGP_REQUIRE(part >= 0 && part <= 2, SVGP_ERR_INVALID, "forward part %d (0, 1 or 2)", part);
    return svgp_big_factor_fwd(c, wl, ws, stream, l0, nl, part);
}
extern "C" int svgp_gp_factor_bwd_channels_part(const svgp_mnist_cfg* c, int l0, int nl, int part, double* ws,
                                                const double* state, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX, SVGP_ERR_UNSUPPORTED, "channel windows exist for the large-m path (m > %d)", SVGP_M_MAX);
    SVGP_REQUIRE(l0 >= 0 && nl >= 1 && l0 + nl <= c->L, SVGP_ERR_INVALID, "channel window [%d, %d) outside 0..%d", l0, l0 + nl, c->L);
    SVGP_REQUIRE(part >= 0 && part <= 4, SVGP_ERR_INVALID, "reverse part %d (0..4)", part);
    return svgp_big_factor_bwd(c, wl, ws, state, stream, l0, nl, part);
}

static int factor_fwd_impl(const svgp_mnist_cfg* c, double* ws, int defer_aji, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws, SVGP_ERR_INVALID, "NULL device pointer");
    if (c->m > SVGP_M_MAX) return svgp_big_factor_fwd(c, wl, ws, stream, 0, c->L, defer_aji ? 1 : 0);
    FactArgs a;
    a.defer_aji = defer_aji; a.kl_form = c->kl_form; a.P = svgp_stat_parts(c);
    a.b = c->b; a.m = c->m; a.L = c->L; a.c = c->N_train / (double)c->b_global; a.jitter = c->jitter;
    a.K = ws + wl.K; a.Ki = ws + wl.Ki; a.ldK = ws + wl.ldK; a.S = ws + wl.S; a.v = ws + wl.v; a.Kn = ws + wl.Kn;
    a.Si = ws + wl.Si; a.t = ws + wl.t; a.G = ws + wl.G; a.A = ws + wl.A; a.Aji = ws + wl.Aji; a.mu = ws + wl.mu_hat;
    a.u = ws + wl.u; a.M2 = ws + wl.M2; a.KL = ws + wl.KL; a.q = ws + wl.q;
    const int m = c->m, RB = rows_per_block(m);
    const size_t lds = mat_lds_pad(m, 4) + (size_t)(3 * m + 16 + SVGP_BLOCK) * sizeof(real);
    int rc = m == 32 ? set_dyn_lds(k_gp_factor_fwd<32>, lds) : set_dyn_lds(k_gp_factor_fwd<0>, lds);
    if (rc) return rc;
    const dim3 grid(c->L + (c->b + RB - 1) / RB);
    if (m == 32) hipLaunchKernelGGL(k_gp_factor_fwd<32>, grid, dim3(SVGP_BLOCK), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_gp_factor_fwd<0>, grid, dim3(SVGP_BLOCK), lds, (hipStream_t)stream, a);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

static int posterior_fwd_impl(const svgp_mnist_cfg* c, const double* eps, double* ws, double* state, bool with_aji,
                              void* stream);
extern "C" int svgp_gp_posterior_fwd(const svgp_mnist_cfg* c, const double* eps, double* ws, double* state,
                                     void* stream) {
    return posterior_fwd_impl(c, eps, ws, state, false, stream);
}
// training-phase form (m <= 64; for larger m the plain stage): L more workgroups finish what svgp_gp_factor_fwd_defer_aji
// left out; svgp_gp_stats_bwd (plain) follows later in the step.
extern "C" int svgp_gp_posterior_fwd_with_aji(const svgp_mnist_cfg* c, const double* eps, double* ws, double* state,
                                              void* stream) {
    return posterior_fwd_impl(c, eps, ws, state, true, stream);
}
static int posterior_fwd_impl(const svgp_mnist_cfg* c, const double* eps, double* ws, double* state, bool with_aji,
                              void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    if (c->m > SVGP_M_MAX) return svgp_big_posterior_fwd(c, wl, eps, ws, state, stream);
    PostArgs a;
    a.b = c->b; a.m = c->m; a.L = c->L; a.clip_pv = c->clip_pv; a.c = c->N_train / (double)c->b_global;
    a.use_rng = eps == nullptr;
    a.Kn = ws + wl.Kn; a.knn = ws + wl.knn; a.q = ws + wl.q; a.y = ws + wl.qnet_mu; a.s2 = ws + wl.qnet_var;
    a.Si = ws + wl.Si; a.M2 = ws + wl.M2; a.t = ws + wl.t; a.u = ws + wl.u; a.eps_in = eps; a.state = state;
    a.eps = ws + wl.eps; a.p_m = ws + wl.p_m; a.p_v = ws + wl.p_v; a.e = ws + wl.e; a.d = ws + wl.d; a.z = ws + wl.z;
    a.part = ws + wl.part_sums + (size_t)svgp_n_part(c) * 4;
    const int m = c->m, RB = rows_per_block(m), nb = (c->b + RB - 1) / RB;
    SVGP_REQUIRE((int64_t)c->L * nb <= wl.n_post, SVGP_ERR_INVALID, "partial-sum layout mismatch");
    size_t lds = mat_lds(m, 2) + (size_t)(2 * m + SVGP_BLOCK + 4 * SVGP_BLOCK + 16) * sizeof(real);
    const size_t lds_inv = mat_lds(m, 1) + (size_t)(5 * m + 80) * sizeof(real);
    if (with_aji && lds_inv > lds) lds = lds_inv;
    a.nb = nb; a.with_aji = with_aji; a.jitter = c->jitter; a.Ahat = ws + wl.A; a.Aji = ws + wl.Aji; a.KL = ws + wl.KL;
    LAUNCH_MC(k_gp_posterior_fwd, m, dim3(nb + (with_aji ? 1 : 0), c->L), lds, stream, a);
    return SVGP_OK;
}

static FactBwdArgs make_fb(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, double* ws, const double* state) {
    FactBwdArgs a;
    a.b_global = c->b_global; a.m = c->m; a.L = c->L; a.geco = SVGP_LOSS_FLAGS(c); a.kl_form = c->kl_form; a.P = svgp_stat_parts(c); a.c = c->N_train / (double)c->b_global;
    a.N_train = c->N_train; a.state = state;
    a.K = ws + wl.K; a.Ki = ws + wl.Ki; a.S = ws + wl.S; a.v = ws + wl.v; a.Si = ws + wl.Si; a.t = ws + wl.t;
    a.G = ws + wl.G; a.A = ws + wl.A; a.Aji = ws + wl.Aji; a.mu = ws + wl.mu_hat; a.u = ws + wl.u; a.M2 = ws + wl.M2;
    a.A2 = ws + wl.A2; a.ud = ws + wl.ud; a.td = ws + wl.td;
    a.Kbar_part = ws + wl.fb_part;            // scratch (L,m,m)
    a.Kibar_part = ws + wl.fb_part + (size_t)c->L * c->m * c->m;
    a.vbar = ws + wl.vbar; a.Ssym = ws + wl.Ssym; a.Qm = ws + wl.Qm; a.Kbar = ws + wl.Kbar;
    a.n_riders = 0;
    memset(&a.wg, 0, sizeof(a.wg));
    a.n_stat = 0; a.flags = nullptr;
    memset(&a.st, 0, sizeof(a.st));
    return a;
}

static int factor_bwd_impl(const svgp_mnist_cfg* c, double* ws, const double* state, bool with_final, void* stream,
                           const double* images_for_wgrad = nullptr, bool with_stats = false);
extern "C" int svgp_gp_factor_bwd(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    return factor_bwd_impl(c, ws, state, true, stream);
}
// training-phase pair (m <= 64): the channel sum Kbar is formed by extra workgroups of the posterior reverse launch
extern "C" int svgp_gp_factor_bwd_nofinal(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    return factor_bwd_impl(c, ws, state, false, stream);
}
// ... and the decoder's weight gradients (svgp_mnist_decoder_bwd_weights, 256 threads) ride in the same launch
extern "C" int svgp_gp_factor_bwd_nofinal_wgrad(const svgp_mnist_cfg* c, const double* images, double* ws, const double* state,
                                                void* stream) {
    SVGP_REQUIRE(c && c->m <= SVGP_M_MAX, SVGP_ERR_UNSUPPORTED,
                 "the weight-gradient riders exist for the LDS-resident reverse factor stage (m <= %d)", SVGP_M_MAX);
    SVGP_REQUIRE(images, SVGP_ERR_INVALID, "NULL device pointer");
    return factor_bwd_impl(c, ws, state, false, stream, images);
}
// ... and, when nothing is exchanged between the reverse statistics and the reverse factor stage (one GPU), svgp_gp_stats_bwd as
// P L leading workgroups of that launch too: channel l starts when its own P row partials are there (ws.flags[8 + l])
extern "C" int svgp_gp_stats_factor_bwd_wgrad(const svgp_mnist_cfg* c, const double* images, double* ws, const double* state,
                                              void* stream) {
    SVGP_REQUIRE(c && c->m <= SVGP_M_MAX, SVGP_ERR_UNSUPPORTED,
                 "the merged reverse statistics + factor launch exists for the LDS-resident stage (m <= %d)", SVGP_M_MAX);
    SVGP_REQUIRE(c->L <= 56, SVGP_ERR_UNSUPPORTED, "ws.flags holds 56 channel counters (L = %d)", c->L);
    SVGP_REQUIRE(images, SVGP_ERR_INVALID, "NULL device pointer");
    return factor_bwd_impl(c, ws, state, false, stream, images, true);
}
// m > 64: the two halves of svgp_gp_factor_bwd (gp_large.hip svgp_big_factor_bwd).  _early needs only forward quantities, the
// loss seeds in `state` and (A_hat + jI)^-1: it may run on another stream, ordered after svgp_gp_factor_fwd_aji_tail, beside
// the row stage, the networks and the reverse statistics; _late follows svgp_gp_stats_bwd (and its exchange) and _early.
extern "C" int svgp_gp_factor_bwd_early(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX, SVGP_ERR_UNSUPPORTED, "the split reverse factor stage exists for m > %d", SVGP_M_MAX);
    return svgp_big_factor_bwd(c, wl, ws, state, stream, 0, c->L, 1);
}
// _early in two parts: _early_a (T1, Ki S Ki, T1 A_hat) needs no (A_hat + jI)^-1 and may run BESIDE svgp_gp_factor_fwd_aji_tail
// (a third stream); _early_b (Abar, Gbar, Z, Gbar K) needs both.  The gp_large.hip fb_part hazard: _early_b overwrites the trace
// partials the tail's last kernel reads -- it is ordered behind the tail anyway (it needs (A_hat + jI)^-1).
extern "C" int svgp_gp_factor_bwd_early_a(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX, SVGP_ERR_UNSUPPORTED, "the split reverse factor stage exists for m > %d", SVGP_M_MAX);
    return svgp_big_factor_bwd(c, wl, ws, state, stream, 0, c->L, 3);
}
extern "C" int svgp_gp_factor_bwd_early_b(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX, SVGP_ERR_UNSUPPORTED, "the split reverse factor stage exists for m > %d", SVGP_M_MAX);
    return svgp_big_factor_bwd(c, wl, ws, state, stream, 0, c->L, 4);
}
// _late = _late_a + _late_b: _late_a reads nothing the early half writes and may be issued BEFORE the caller's stream joins the branch
// the early half runs on; _late_b follows the join
extern "C" int svgp_gp_factor_bwd_late_a(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX, SVGP_ERR_UNSUPPORTED, "the split reverse factor stage exists for m > %d", SVGP_M_MAX);
    return svgp_big_factor_bwd(c, wl, ws, state, stream, 0, c->L, 6);
}
extern "C" int svgp_gp_factor_bwd_late_b(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX, SVGP_ERR_UNSUPPORTED, "the split reverse factor stage exists for m > %d", SVGP_M_MAX);
    return svgp_big_factor_bwd(c, wl, ws, state, stream, 0, c->L, 7);
}
// _late_b = _late_b_channels (the channel block: X sandwiches, Ssym, Sgs) + _late_b_kbar (the single-matrix chain of the gradient of Ki:
// five small launches that read nothing of the channel block and may run beside it on another stream) + _late_b_final (round 6)
extern "C" int svgp_gp_factor_bwd_late_b_channels(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX, SVGP_ERR_UNSUPPORTED, "the split reverse factor stage exists for m > %d", SVGP_M_MAX);
    return svgp_big_factor_bwd(c, wl, ws, state, stream, 0, c->L, 8);
}
extern "C" int svgp_gp_factor_bwd_late_b_kbar(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX, SVGP_ERR_UNSUPPORTED, "the split reverse factor stage exists for m > %d", SVGP_M_MAX);
    return svgp_big_factor_bwd(c, wl, ws, state, stream, 0, c->L, 9);
}
extern "C" int svgp_gp_factor_bwd_late_b_final(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX, SVGP_ERR_UNSUPPORTED, "the split reverse factor stage exists for m > %d", SVGP_M_MAX);
    return svgp_big_factor_bwd(c, wl, ws, state, stream, 0, c->L, 10);
}
extern "C" int svgp_gp_factor_bwd_late(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(c->m > SVGP_M_MAX, SVGP_ERR_UNSUPPORTED, "the split reverse factor stage exists for m > %d", SVGP_M_MAX);
    return svgp_big_factor_bwd(c, wl, ws, state, stream, 0, c->L, 2);
}
static int factor_bwd_impl(const svgp_mnist_cfg* c, double* ws, const double* state, bool with_final, void* stream,
                           const double* images_for_wgrad, bool with_stats) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    if (c->m > SVGP_M_MAX) return svgp_big_factor_bwd(c, wl, ws, state, stream, 0, c->L);
    FactBwdArgs a = make_fb(c, wl, ws, state);
    const int m = c->m;
    const int keep = m * m <= 4 * SVGP_BLOCK ? 1 : 0;          // + the LDS accumulator of Kbar_l
    size_t lds = mat_lds_pad(m, 4 + keep) + (size_t)(6 * m) * sizeof(real), lds_w = 0;
    bool five = false;
    if (images_for_wgrad) {
        static const int n_types = [] { const char* e = getenv("SVGP_DEC_RIDER_TYPES"); return (e && e[0] >= '1' && e[0] <= '3') ? e[0] - '0' : 3; }();
        a.wg = svgp_make_dec_wgrad_args(c, wl, images_for_wgrad, ws, state, n_types);
        a.n_riders = a.wg.n_slots * a.wg.n_types;
        lds_w = (size_t)svgp_vae::dec_wgrad_lds(SVGP_BLOCK, c->L, n_types) * sizeof(real);
        if (lds_w > lds) lds = lds_w;
    }
    if (with_stats) {
        size_t lds_s = 0;
        a.st = make_stat_args(c, wl, ws, state, 1, 0, &lds_s);
        a.n_stat = a.P * c->L;
        a.wait_n = a.P;
        // (SVGP_STAT_FOUR=1: the four-matrix form also where five fit -- what 32 < m <= 64 runs; tests compare the two)
        five = !c->kl_form && m * m <= 4 * SVGP_BLOCK && !getenv("SVGP_STAT_FOUR");
        if (five) lds = mat_lds_pad(m, 5 + keep) + (size_t)(6 * m) * sizeof(real);
        if (images_for_wgrad && lds_w > lds) lds = lds_w;
        a.flags = reinterpret_cast<unsigned long long*>(ws + wl.flags);
        if (lds_s > lds) lds = lds_s;
    }
    const dim3 grid(a.n_stat + c->L + a.n_riders);
    int rc;
    if (with_stats) {
        if (five) {
            rc = m == 32 ? set_dyn_lds(k_gp_factor_bwd<32, 2>, lds) : set_dyn_lds(k_gp_factor_bwd<0, 2>, lds);
            if (rc) return rc;
            if (m == 32) hipLaunchKernelGGL((k_gp_factor_bwd<32, 2>), grid, dim3(SVGP_BLOCK), lds, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((k_gp_factor_bwd<0, 2>), grid, dim3(SVGP_BLOCK), lds, (hipStream_t)stream, a);
        } else {
            rc = m == 32 ? set_dyn_lds(k_gp_factor_bwd<32, 1>, lds) : set_dyn_lds(k_gp_factor_bwd<0, 1>, lds);
            if (rc) return rc;
            if (m == 32) hipLaunchKernelGGL((k_gp_factor_bwd<32, 1>), grid, dim3(SVGP_BLOCK), lds, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((k_gp_factor_bwd<0, 1>), grid, dim3(SVGP_BLOCK), lds, (hipStream_t)stream, a);
        }
    } else {
        rc = m == 32 ? set_dyn_lds(k_gp_factor_bwd<32>, lds) : set_dyn_lds(k_gp_factor_bwd<0>, lds);
        if (rc) return rc;
        if (m == 32) hipLaunchKernelGGL(k_gp_factor_bwd<32>, grid, dim3(SVGP_BLOCK), lds, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(k_gp_factor_bwd<0>, grid, dim3(SVGP_BLOCK), lds, (hipStream_t)stream, a);
    }
    SVGP_LAUNCH_CHECK();
    if (with_final) {
        hipLaunchKernelGGL(k_gp_factor_bwd_final, dim3((m * m + SVGP_BLOCK - 1) / SVGP_BLOCK), dim3(SVGP_BLOCK), 0,
                           (hipStream_t)stream, a);
        SVGP_LAUNCH_CHECK();
    }
    return SVGP_OK;
}

static int posterior_bwd_impl(const svgp_mnist_cfg* c, double* ws, const double* state, bool with_final, void* stream, int pass = 0);
extern "C" int svgp_gp_posterior_bwd(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    return posterior_bwd_impl(c, ws, state, false, stream);
}
extern "C" int svgp_gp_posterior_bwd_with_final(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    return posterior_bwd_impl(c, ws, state, true, stream);
}
static PostBwdArgs make_pb(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, double* ws, const double* state, bool with_final) {
    PostBwdArgs a;
    a.b = c->b; a.m = c->m; a.L = c->L; a.geco = SVGP_LOSS_FLAGS(c); a.c = c->N_train / (double)c->b_global; a.state = state;
    a.Kn = ws + wl.Kn; a.y = ws + wl.qnet_mu; a.s2 = ws + wl.qnet_var; a.p_m = ws + wl.p_m; a.p_v = ws + wl.p_v;
    a.e = ws + wl.e; a.d = ws + wl.d; a.g_pv = ws + wl.g_pv; a.g_pm = ws + wl.g_pm; a.mvbar = ws + wl.mvbar;
    a.Si = ws + wl.Si; a.Qm = ws + wl.Qm; a.Ssym = ws + wl.Ssym; a.u = ws + wl.u; a.t = ws + wl.t; a.vbar = ws + wl.vbar;
    a.Ki = ws + wl.Ki;
    a.Knbar_part = ws + wl.Knbar_part;
    a.Knbar = ws + wl.Knbar; a.knnbar = ws + wl.knnbar; a.ybar = ws + wl.ybar; a.s2bar = ws + wl.s2bar;
    const int m = c->m, RB = rows_per_block(m), nb = (c->b + RB - 1) / RB;
    a.nb_rows = nb; a.n_final = with_final ? (m * m + SVGP_BLOCK - 1) / SVGP_BLOCK : 0;
    a.b_global = c->b_global; a.N_train = c->N_train;
    a.Kbar_part = ws + wl.fb_part; a.Kbar = ws + wl.Kbar;
    return a;
}
// pass: 0 = both passes, 1 = the per-channel row terms only (ybar, s2bar, the (L, b, m) partials of Knbar)
static int posterior_bwd_impl(const svgp_mnist_cfg* c, double* ws, const double* state, bool with_final, void* stream, int pass) {
    GET_LAYOUTS();
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    if (c->m > SVGP_M_MAX) return svgp_big_posterior_bwd(c, wl, ws, state, stream);
    PostBwdArgs a = make_pb(c, wl, ws, state, with_final);
    const int m = c->m, nb = a.nb_rows;
    const size_t lds = mat_lds(m, 3) + (size_t)(3 * m + SVGP_BLOCK + 2 * SVGP_BLOCK) * sizeof(real);
    LAUNCH_MC(k_gp_posterior_bwd_l, m, dim3(nb, c->L), lds, stream, a);
    if (pass == 1) return SVGP_OK;
    const size_t lds2 = mat_lds(m, 1) + (size_t)(SVGP_BLOCK + (SVGP_BLOCK / m) * (1 + c->L)) * sizeof(real);
    LAUNCH_MC(k_gp_posterior_bwd_sum, m, dim3(nb + a.n_final), lds2, stream, a);
    return SVGP_OK;
}
// m <= 64, training step: pass 1 of the reverse row stage alone; its pass 2 (the sums over channels) then rides in
// svgp_mnist_encoder_bwd_km_sum
extern "C" int svgp_gp_posterior_bwd_rows(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    SVGP_REQUIRE(c && c->m <= SVGP_M_MAX, SVGP_ERR_UNSUPPORTED, "the two-pass reverse row stage exists for m <= %d", SVGP_M_MAX);
    return posterior_bwd_impl(c, ws, state, true, stream, 1);
}
