// Inverse of ONE general square matrix by LU with partial pivoting: the reference's `tf.linalg.inv(K_mm)` WITHOUT jitter in the
// SPRITES conditional-generation path (SPRITES_experiment.py:178, consumed at SVGPVAE_model.py:610-635).  With the linear x linear
// kernels K_mm (m = 800) has rank <= L_action * L_character = 128 < m: a Cholesky / no-pivot elimination -- the library's SPD
// inverse -- has no answer there, the row-pivoted LU returns the (huge but finite) matrix the reference works with.
//
// Blocked right-looking getrf, panels of 16 columns (two rows x 16 columns = 64 registers per thread of the 1024):
//   k_lu_panel       ONE workgroup of 1024 threads holds the (m - k0) x 16 panel in registers (two rows per thread, m <= 2048);
//                    per column: arg-max |a| over the rows (wave shuffles + one LDS round; ties -> the lowest row, as idamax),
//                    row swap and pivot-row broadcast through LDS, multipliers by IEEE division, rank-1 update in registers.
//   k_lu_swap_solve  one thread per column outside the panel: the panel's 16 row swaps in order; right of the panel also
//                    the unit-lower solve U12 = L11^-1 A12 (L11 in LDS).
//   trailing update  A22 -= L21 U12: the batched f64 MFMA GEMM of linalg.hip with K = 16.
// Inverse: X = U^-1 L^-1 P by two svgp_trsm_batched calls (left, lower; U enters as the lower-triangular U^T with trans = 1).
// Not a hot path (once per evaluation, ~1 ms at m = 800): written for correctness and LAPACK's pivoting rule, not for speed.
#include "common.hpp"

#define LU_NB 16
#define LU_NT 1024
#define LU_RPT 2           // rows per thread of the panel kernel: m <= LU_NT * LU_RPT
#define LU_MAX_M (LU_NT * LU_RPT)

namespace {

__global__ __launch_bounds__(LU_NT) void k_lu_panel(int m, int k0, real* __restrict__ A, int* __restrict__ ipiv) {
    __shared__ real prow[LU_NB];
    __shared__ real srow[2][LU_NB];
    __shared__ real wval[LU_NT / 64];
    __shared__ int widx[LU_NT / 64];
    __shared__ int spiv;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nrows = m - k0, nb = nrows < LU_NB ? nrows : LU_NB;
    real a[LU_RPT][LU_NB];
#pragma unroll
    for (int q = 0; q < LU_RPT; ++q) {
        const int row = t + q * LU_NT;
#pragma unroll
        for (int c = 0; c < LU_NB; ++c) a[q][c] = (row < nrows && c < nb) ? A[(size_t)(k0 + row) * m + k0 + c] : real(0);
    }
#pragma unroll
    for (int j = 0; j < LU_NB; ++j) {
        if (j < nb) {                                              // (uniform)
            // ---- 1. pivot = first row >= j with the largest |a[.][j]|
            real best = real(-1);
            int bi = 0x7fffffff;
#pragma unroll
            for (int q = 0; q < LU_RPT; ++q) {
                const int row = t + q * LU_NT;
                const real v = fabs(a[q][j]);
                if (row >= j && row < nrows && (v > best || (v == best && row < bi))) { best = v; bi = row; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const real ov = __shfl_down(best, o, 64);
                const int oi = __shfl_down(bi, o, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (lane == 0) { wval[wave] = best; widx[wave] = bi; }
            __syncthreads();
            if (t == 0) {
                real bv = wval[0];
                int bx = widx[0];
                for (int w = 1; w < LU_NT / 64; ++w)
                    if (wval[w] > bv || (wval[w] == bv && widx[w] < bx)) { bv = wval[w]; bx = widx[w]; }
                if (bx == 0x7fffffff) bx = j;                      // (a column of NaNs: keep the diagonal row)
                spiv = bx;
                ipiv[k0 + j] = k0 + bx;
            }
            __syncthreads();
            const int piv = spiv;
            // ---- 2. swap rows j and piv inside the panel (through LDS: the two rows live in different threads)
            if (piv != j) {
#pragma unroll
                for (int q = 0; q < LU_RPT; ++q) {
                    const int row = t + q * LU_NT;
                    if (row == j || row == piv) {
#pragma unroll
                        for (int c = 0; c < LU_NB; ++c) srow[row == j ? 0 : 1][c] = a[q][c];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < LU_RPT; ++q) {
                    const int row = t + q * LU_NT;
                    if (row == j || row == piv) {
#pragma unroll
                        for (int c = 0; c < LU_NB; ++c) a[q][c] = srow[row == j ? 1 : 0][c];
                    }
                }
            }
            // ---- 3. broadcast the pivot row
#pragma unroll
            for (int q = 0; q < LU_RPT; ++q)
                if (t + q * LU_NT == j) {
#pragma unroll
                    for (int c = 0; c < LU_NB; ++c) prow[c] = a[q][c];
                }
            __syncthreads();
            // ---- 4. multipliers + rank-1 update of the columns right of j (a zero pivot leaves the column as it is: LAPACK's info > 0)
            const real pv = prow[j];
            if (pv != real(0)) {
#pragma unroll
                for (int q = 0; q < LU_RPT; ++q) {
                    const int row = t + q * LU_NT;
                    if (row > j && row < nrows) {
                        const real l = a[q][j] / pv;
                        a[q][j] = l;
#pragma unroll
                        for (int c = j + 1; c < LU_NB; ++c) a[q][c] = fma(-l, prow[c], a[q][c]);
                    }
                }
            }
            __syncthreads();                                       // prow / srow / spiv are rewritten by the next column
        }
    }
#pragma unroll
    for (int q = 0; q < LU_RPT; ++q) {
        const int row = t + q * LU_NT;
        if (row < nrows) {
#pragma unroll
            for (int c = 0; c < LU_NB; ++c)
                if (c < nb) A[(size_t)(k0 + row) * m + k0 + c] = a[q][c];
        }
    }
}

// one thread per column c outside the panel [k0, k0 + nb): the panel's row swaps; right of the panel also x <- L11^-1 x
__global__ __launch_bounds__(256) void k_lu_swap_solve(int m, int k0, int nb, real* __restrict__ A, const int* __restrict__ ipiv) {
    __shared__ real L11[LU_NB][LU_NB + 1];
    __shared__ int piv[LU_NB];
    for (int o = threadIdx.x; o < LU_NB * LU_NB; o += blockDim.x) {
        const int i = o / LU_NB, j = o % LU_NB;
        L11[i][j] = (i < nb && j < i) ? A[(size_t)(k0 + i) * m + k0 + j] : real(0);
    }
    if ((int)threadIdx.x < LU_NB) piv[threadIdx.x] = (int)threadIdx.x < nb ? ipiv[k0 + threadIdx.x] : 0;
    __syncthreads();
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // 0 .. m - nb - 1
    if (idx >= m - nb) return;
    const int c = idx < k0 ? idx : idx + nb;
    for (int j = 0; j < nb; ++j) {
        const int p = piv[j];
        if (p != k0 + j) {
            const real u = A[(size_t)(k0 + j) * m + c], v = A[(size_t)p * m + c];
            A[(size_t)(k0 + j) * m + c] = v;
            A[(size_t)p * m + c] = u;
        }
    }
    if (c < k0) return;
    real x[LU_NB];
#pragma unroll
    for (int i = 0; i < LU_NB; ++i) x[i] = i < nb ? A[(size_t)(k0 + i) * m + c] : real(0);
#pragma unroll
    for (int i = 1; i < LU_NB; ++i) {
        real s = x[i];
#pragma unroll
        for (int j = 0; j < i; ++j) s = fma(-L11[i][j], x[j], s);
        x[i] = s;
    }
#pragma unroll
    for (int i = 0; i < LU_NB; ++i)
        if (i < nb) A[(size_t)(k0 + i) * m + c] = x[i];
}

__global__ void k_lu_perm(int m, const int* __restrict__ ipiv, int* __restrict__ perm) {
    if (blockIdx.x || threadIdx.x) return;
    for (int i = 0; i < m; ++i) perm[i] = i;
    for (int k = 0; k < m; ++k) {
        const int p = ipiv[k], u = perm[k];
        perm[k] = perm[p];
        perm[p] = u;
    }
}

// Lm = unit-lower L, UT = U^T (lower), X = P (row i = e_perm[i]): the right-hand side of L U X = P
__global__ __launch_bounds__(256) void k_lu_split(int m, const real* __restrict__ LU, const int* __restrict__ perm,
                                                  real* __restrict__ Lm, real* __restrict__ UT, real* __restrict__ X) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (long long)m * m) return;
    const int i = (int)(o / m), j = (int)(o % m);
    Lm[o] = i > j ? LU[o] : (i == j ? real(1) : real(0));
    UT[o] = i >= j ? LU[(size_t)j * m + i] : real(0);
    X[o] = perm[i] == j ? real(1) : real(0);
}

}  // namespace

extern "C" size_t svgp_lu_inverse_workspace_elems(int m) {
    if (m < 1) return 0;
    const size_t mm = (size_t)m * m;
    return 3 * mm + 2 * (((size_t)m + 1) / 2 + 8) + svgp_trsm_workspace_elems(m, m, 1);
}

// A (m x m, row-major, contiguous; not modified) -> Ainv (m x m) = A^-1 by LU with partial pivoting.  A and Ainv may not alias.
extern "C" int svgp_lu_inverse(int m, const double* A, double* Ainv, double* work, void* stream) {
    SVGP_REQUIRE(m >= 1 && m <= LU_MAX_M, SVGP_ERR_UNSUPPORTED, "svgp_lu_inverse: m=%d outside 1..%d", m, LU_MAX_M);
    SVGP_REQUIRE(A && Ainv && work && A != Ainv, SVGP_ERR_INVALID, "NULL device pointer / aliased output");
    hipStream_t s = (hipStream_t)stream;
    const size_t mm = (size_t)m * m;
    real* LU = work;
    real* Lm = LU + mm;
    real* UT = Lm + mm;
    int* ipiv = reinterpret_cast<int*>(UT + mm);
    int* perm = ipiv + 2 * (((size_t)m + 1) / 2 + 8);       // (int view of the second integer block)
    real* twork = UT + mm + 2 * (((size_t)m + 1) / 2 + 8);
    SVGP_CHECK_HIP(hipMemcpyAsync(LU, A, mm * sizeof(real), hipMemcpyDeviceToDevice, s));
    for (int k0 = 0; k0 < m; k0 += LU_NB) {
        const int nb = m - k0 < LU_NB ? m - k0 : LU_NB, rest = m - k0 - nb;
        hipLaunchKernelGGL(k_lu_panel, dim3(1), dim3(LU_NT), 0, s, m, k0, LU, ipiv);
        SVGP_LAUNCH_CHECK();
        if (m - nb > 0) {
            hipLaunchKernelGGL(k_lu_swap_solve, dim3((m - nb + 255) / 256), dim3(256), 0, s, m, k0, nb, LU, ipiv);
            SVGP_LAUNCH_CHECK();
        }
        if (rest > 0) {
            int rc = svgp_dgemm_batched(0, 0, rest, rest, nb, -1.0, LU + (size_t)(k0 + nb) * m + k0, m, 0,
                                        LU + (size_t)k0 * m + k0 + nb, m, 0, 1.0, LU + (size_t)(k0 + nb) * m + k0 + nb, m, 0, 1,
                                        stream);
            if (rc) return rc;
        }
    }
    hipLaunchKernelGGL(k_lu_perm, dim3(1), dim3(64), 0, s, m, ipiv, perm);
    SVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_lu_split, dim3((unsigned)((mm + 255) / 256)), dim3(256), 0, s, m, LU, perm, Lm, UT, Ainv);
    SVGP_LAUNCH_CHECK();
    int rc = svgp_trsm_batched(0, 0, m, m, Lm, m, 0, Ainv, m, 0, 1, twork, stream);       // Y = L^-1 P
    if (rc) return rc;
    return svgp_trsm_batched(0, 1, m, m, UT, m, 0, Ainv, m, 0, 1, twork, stream);         // X = U^-1 Y  (U = (U^T)^T)
}
