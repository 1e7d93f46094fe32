// Batched float64 Cholesky family on device matrices (north_star: "Cholesky of K_mm and the triangular solves"):
//   svgp_potrf_batched   A = L L^T, lower, blocked right-looking; log det = 2 sum log diag(L)
//   svgp_trsm_batched    op(L) X = B  /  X op(L) = B
//   svgp_potri_batched   A^-1 = L^-T L^-1 from the factor (triangular inverse by recursive halving + L^-T L^-1 product)
// They replace tf.linalg.cholesky + log(diag_part) (SVGPVAE_model.py:270-274), tf.linalg.inv of the SPD matrices
// (:239,319,331) and the solves behind them.  Blocked with 64 x 64 diagonal blocks: one workgroup factors a diagonal
// block in LDS and inverts its triangular factor; everything else -- panel solve, trailing update, the products of the
// triangular inverse and of L^-T L^-1 -- is the batched f64 MFMA GEMM of linalg.hip with triangular structure hints
// (tiles above the diagonal and k-panels where an operand is structurally zero are skipped).
#include "common.hpp"

#define CB 64          // diagonal block
#define CLD (CB + 1)

namespace {

__device__ __forceinline__ real wave_sum_c(real x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}

// Lower Cholesky of the 64 x 64 block in Ls (identity-padded beyond n; 256 threads), unblocked right-looking with the
// matrix in REGISTERS: thread (ti = tid >> 4, tj = tid & 15) owns the 16 elements (ti + 16 a, tj + 16 b).  Per column k the
// owners of that column publish it (64 values) through a double-buffered LDS vector -- one barrier per column and 16 FMAs
// per thread; the first version kept the matrix in LDS (3 barriers, 4 LDS accesses per update, an integer division per
// element): 113 us per diagonal block at m = 800 against the GEMMs' 40-90 us per block step.
// Ls lower triangle <- L, strict upper triangle <- 0.  A non-positive pivot yields NaN (sqrt), as LAPACK's failure would.
// 1 / sqrt(x): v_rsq_f64 (about 2^-26) + two Newton steps y <- y + y/2 (1 - x y^2); sqrt(x) = x * (1 / sqrt(x)).  The library
// sqrt() and the IEEE division each expand to a chain of ~25 dependent float64 instructions, and one of each sat on the
// critical path of every column of the diagonal-block factorisation.  A non-positive x gives NaN / inf, as sqrt would.
__device__ __forceinline__ real fast_rsqrt(real x) {
    real y = __builtin_amdgcn_rsq(x);
    real e = fma(-x * y, y, real(1));
    y = fma(real(0.5) * y, e, y);
    e = fma(-x * y, y, real(1));
    y = fma(real(0.5) * y, e, y);
    return y;
}
// rdiag[k] receives 1 / L_kk (the triangular inverse multiplies by it instead of dividing)
// (Loading the thread's 16 elements straight from global memory instead of through the LDS image was measured slower:
// 79 against 58 us per launch of the diagonal-block kernel at m = 800, batch 65.)
__device__ __forceinline__ void chol64_lds(real (*Ls)[CLD], real (*colk)[CB], real* rdiag) {
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    real r[4][4];
    __syncthreads();                                  // the caller has just filled Ls
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) r[a][b] = Ls[ti + 16 * a][tj + 16 * b];
#pragma unroll
    for (int k = 0; k < CB; ++k) {
        const int kb = k >> 4, kr = k & 15;          // compile-time after unrolling
        real* ck = colk[k & 1];
        if (tj == kr) {
#pragma unroll
            for (int a = 0; a < 4; ++a) ck[ti + 16 * a] = r[a][kb];
        }
        __syncthreads();
        const real rd = fast_rsqrt(ck[k]), sd = ck[k] * rd;
        if (tid == 0) rdiag[k] = rd;
        real li[4], lj[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) li[a] = (ti + 16 * a > k) ? ck[ti + 16 * a] * rd : real(0);
#pragma unroll
        for (int b = 0; b < 4; ++b) lj[b] = (tj + 16 * b > k) ? ck[tj + 16 * b] * rd : real(0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) r[a][b] -= li[a] * lj[b];
        if (tj == kr) {                               // the finished column k
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int i = ti + 16 * a;
                r[a][kb] = i == k ? sd : (i > k ? li[a] : r[a][kb]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = ti + 16 * a, j = tj + 16 * b;
            Ls[i][j] = j <= i ? r[a][b] : real(0);
        }
    __syncthreads();
}

// The same factorisation on ONE wave with a matrix row per lane (64 float64 registers): per column k every lane publishes its
// a_ik through the LDS vector, reads the column back as wave-uniform (broadcast) LDS loads and applies
// a_ij -= (a_ik / a_kk) a_jk to its row -- no workgroup barrier (LDS operations of one wave complete in order), the next
// column's element is updated and published first so its LDS round trip hides behind the rest of the row update.
// 2016 FMAs per lane in all (4 cycles each) against 64 x (barrier + LDS round trip + rsqrt chain + 16 FMAs) of chol64_lds:
// DIAGTIMES in DESIGN.md section 11.  The other three waves wait at the closing barrier.
#define CHOL_PIN8(r, a) asm volatile("" : "+v"(r[a]), "+v"(r[a + 1]), "+v"(r[a + 2]), "+v"(r[a + 3]), "+v"(r[a + 4]), "+v"(r[a + 5]), \
                                        "+v"(r[a + 6]), "+v"(r[a + 7]) :: "memory")
// column k (a template parameter: #pragma unroll gives up on a body this large, and a rolled loop would index the row dynamically).
// A wave issues in order, so the column is software-pipelined by hand: c1 = a_(k+1)k (published by the previous column right
// after it was updated) is loaded by the previous column and arrives in `c1`; the pivot a_kk came from lane k's register
// (v_readlane) and its rsqrt chain ran between the FMAs of the previous column (`ckk`, `rd`).
__device__ __forceinline__ real chol_lane_bcast(real x, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
// fast_rsqrt cut into its dependent stages, so that the caller can place one stage between groups of independent FMAs
struct RsqPipe { real x, y, a, h, e; };
__device__ __forceinline__ void rsq_stage(int st, RsqPipe& p) {
    switch (st) {
        case 0: p.y = __builtin_amdgcn_rsq(p.x); break;
        case 1: case 4: p.a = -p.x * p.y; p.h = real(0.5) * p.y; break;
        case 2: case 5: p.e = fma(p.a, p.y, real(1)); break;
        case 3: case 6: p.y = fma(p.h, p.e, p.y); break;
    }
    asm volatile("" : "+v"(p.y), "+v"(p.a), "+v"(p.h), "+v"(p.e));
}
#define RSQ_STAGES 7
#define CHOL_PIN4(r, a) asm volatile("" : "+v"(r[a]), "+v"(r[a + 1]), "+v"(r[a + 2]), "+v"(r[a + 3]))
template <int k>
__device__ __forceinline__ void chol64_wave_col(real (&r)[CB], real (*colk)[CB], real* rdiag, const int i, const real ckk, const real rd,
                                                const real c1) {
    const real* ck = colk[k & 1];
    const real t = i > k ? r[k] * (rd * rd) : real(0);
    real ckk_n = 0, rd_n = 0, c1_n = 0;
    constexpr int NQ = CB / 16, q0 = (k + 2) >> 4;
    real c[16], cn[16];
    if constexpr (k + 1 < CB) {
        r[k + 1] -= t * c1;
        colk[(k + 1) & 1][i] = r[k + 1];
        asm volatile("" ::: "memory");
        if constexpr (k + 2 < CB) c1_n = colk[(k + 1) & 1][k + 2];      // the next column's c1 = a_(k+2)(k+1), lane k + 2's store above
    }                                                                   // (LDS operations of one wave complete in order)
    if constexpr (q0 < NQ) {
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = ck[16 * q0 + e];
    }
    RsqPipe p = {real(1), real(0), real(0), real(0), real(0)};
    int st = RSQ_STAGES;
    if constexpr (k + 1 < CB) {
        p.x = ckk_n = chol_lane_bcast(r[k + 1], k + 1);
        st = 0;
    }
    // the rest of the row in aligned chunks of 16: the loads of chunk q + 1 are issued before the FMAs of chunk q; the pins keep
    // each chunk's FMAs where they are written (left alone the compiler lets them sink to their uses with every loaded column
    // value live: 8 KB of scratch per lane)
#pragma unroll
    for (int q = q0; q < NQ; ++q) {
        if (q + 1 < NQ) {
#pragma unroll
            for (int e = 0; e < 16; ++e) cn[e] = ck[16 * (q + 1) + e];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (16 * q + e >= k + 2) r[16 * q + e] -= t * c[e];
            if ((e & 3) == 3) {
                if (e == 15) asm volatile("" ::: "memory");
                CHOL_PIN4(r, 16 * q + e - 3);
                if (st < RSQ_STAGES) rsq_stage(st++, p);       // one stage of the next pivot's rsqrt per four FMAs
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = cn[e];
    }
#pragma unroll
    for (int s2 = 0; s2 < RSQ_STAGES; ++s2)
        if (s2 >= st) rsq_stage(s2, p);
    rd_n = p.y;
    r[k] = i == k ? ckk * rd : (i > k ? r[k] * rd : r[k]);
    if (i == 0) rdiag[k] = rd;
    if constexpr (k + 1 < CB) chol64_wave_col<k + 1>(r, colk, rdiag, i, ckk_n, rd_n, c1_n);
}
__device__ __forceinline__ void chol64_wave(real (*Ls)[CLD], real (*colk)[CB], real* rdiag) {
    __syncthreads();                                  // the caller has just filled Ls
    if (threadIdx.x < CB) {
        const int i = threadIdx.x;
        real r[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) r[j] = Ls[i][j];
        colk[0][i] = r[0];
        asm volatile("" ::: "memory");
        const real c1 = colk[0][1], ckk = chol_lane_bcast(r[0], 0);
        chol64_wave_col<0>(r, colk, rdiag, i, ckk, fast_rsqrt(ckk), c1);
#pragma unroll
        for (int j = 0; j < CB; ++j) Ls[i][j] = j <= i ? r[j] : real(0);
    }
    __syncthreads();
}
__device__ __forceinline__ void chol64(real (*Ls)[CLD], real (*colk)[CB], real* rdiag, int onewave) {
    if (onewave) chol64_wave(Ls, colk, rdiag);
    else chol64_lds(Ls, colk, rdiag);
}

// X = L^-1 for the lower-triangular 64 x 64 block in Ls (identity-padded), by halves:
//   X11 = L11^-1 (wave 0), X22 = L22^-1 (wave 1): lane j solves L x = e_j with x in registers, x_i = ((i == j) - sum_{k<i} l_ik x_k) / l_ii
//     -- entries above the diagonal come out as exact zeros, so the code is lane-uniform, the l_ik reads are LDS
//     broadcasts and there is no barrier (32 rows: 496 dependent-chain FMAs per lane instead of the 2016 of a 64-row solve);
//   X21 = -X22 (L21 X11): two 32 x 32 x 32 products on the f64 MFMA, one 16 x 16 tile per wave, through the (zero, unused)
//     upper-right quadrant of Ls as scratch, which is zeroed again.
// In-kernel timestamps of the diagonal-block kernel (m = 800, batch 65; 100 MHz s_memrealtime): block load 2.2 us, Cholesky
// 30 us (475 ns per column: one barrier + an LDS round trip + the rsqrt chain), log det 0.7, this inverse 15.7 -> 4.8 us
// (it was one 64-row solve on one wave), stores 1.4.  Xs gets the full block (zeros above the diagonal).
__device__ __forceinline__ void trinv64_lds(real (*Ls)[CLD], real (*Xs)[CLD], const real* rdiag) {
    typedef double d4c __attribute__((ext_vector_type(4)));
    constexpr int H = CB / 2;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    __syncthreads();
    if (wave < 2 && lane < H) {
        const int o = wave * H, j = lane;
        real x[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            real s0 = 0, s1 = 0, s2 = 0, s3 = 0;      // four partial sums: the chain of dependent FMAs is the cost
#pragma unroll
            for (int k = 0; k + 3 < i; k += 4) {
                s0 += Ls[o + i][o + k] * x[k]; s1 += Ls[o + i][o + k + 1] * x[k + 1];
                s2 += Ls[o + i][o + k + 2] * x[k + 2]; s3 += Ls[o + i][o + k + 3] * x[k + 3];
            }
#pragma unroll
            for (int k = i & ~3; k < i; ++k) s0 += Ls[o + i][o + k] * x[k];
            x[i] = ((i == j ? real(1) : real(0)) - ((s0 + s1) + (s2 + s3))) * rdiag[o + i];
        }
#pragma unroll
        for (int i = 0; i < H; ++i) Xs[o + i][o + j] = x[i];
    } else if (wave >= 2) {                           // upper-right quadrant of X: zeros
        for (int e = tid - 128; e < H * H; e += 128) Xs[e / H][H + e % H] = real(0);
    }
    __syncthreads();
    const int r = lane & 15, q = lane >> 4, ti = 16 * (wave >> 1), tj = 16 * (wave & 1);
    {   // T = L21 X11 -> Ls[0:32][32:64]
        d4c acc = {0, 0, 0, 0};
#pragma unroll
        for (int k0 = 0; k0 < H; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[H + ti + r][k0 + q], Xs[k0 + q][tj + r], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) Ls[ti + q + 4 * e][H + tj + r] = acc[e];
    }
    __syncthreads();
    {   // X21 = -X22 T
        d4c acc = {0, 0, 0, 0};
#pragma unroll
        for (int k0 = 0; k0 < H; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[H + ti + r][H + k0 + q], Ls[k0 + q][H + tj + r], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) Xs[H + ti + q + 4 * e][tj + r] = -acc[e];
    }
    __syncthreads();
    for (int e = tid; e < H * H; e += blockDim.x) Ls[e / H][H + e % H] = real(0);
    __syncthreads();
}

struct DiagArgs {
    int m, r0, nbk, lda, batch, first;     // block rows r0 .. r0 + nbk of an m x m matrix
    long long sA;
    real* A;          // (batch, m, lda): block read, L written back (upper part of the block zeroed)
    real* Linv;       // (batch, nblk, 64, 64): slot r0 / 64 receives L_kk^-1 (identity-padded)
    int nblk;
    real* logdet;     // (batch): += 2 sum log diag  (= when first)
    int onewave;      // chol64_wave instead of chol64_lds
};
// one workgroup per matrix: Cholesky of the diagonal block + inverse of its factor + log det contribution
// dynamic LDS (DIAG_LDS_BYTES: two padded 64 x 64 blocks + the column buffer = 67.6 KB, above the 64 KB a kernel gets
// without hipFuncAttributeMaxDynamicSharedMemorySize)
#define DIAG_LDS_BYTES ((2 * CB * CLD + 3 * CB) * sizeof(real))
__global__ __launch_bounds__(256) void k_chol_diag(DiagArgs g) {
    extern __shared__ __align__(16) unsigned char diag_lds[];
    // column buffer first: its addresses stay below 64 KB, the range of a DS instruction's immediate offset
    real (*colk)[CB] = reinterpret_cast<real (*)[CB]>(diag_lds);
    real* rdiag = reinterpret_cast<real*>(diag_lds + 2 * CB * sizeof(real));
    real (*Ls)[CLD] = reinterpret_cast<real (*)[CLD]>(diag_lds + 3 * CB * sizeof(real));
    real (*Xs)[CLD] = reinterpret_cast<real (*)[CLD]>(diag_lds + (3 * CB + CB * CLD) * sizeof(real));
    const int l = blockIdx.x, n = g.nbk;
    real* A = g.A + (size_t)l * g.sA + (size_t)g.r0 * g.lda + g.r0;
    for (int e = threadIdx.x; e < CB * CB; e += blockDim.x) {
        const int i = e / CB, j = e % CB;
        Ls[i][j] = (i < n && j < n) ? A[(size_t)i * g.lda + j] : (i == j ? real(1) : real(0));     // identity pad
    }
    chol64(Ls, colk, rdiag, g.onewave);
    real lg = (threadIdx.x < n) ? log(Ls[threadIdx.x][threadIdx.x]) : real(0);
    lg = wave_sum_c(lg);
    if (threadIdx.x == 0) g.logdet[l] = (g.first ? real(0) : g.logdet[l]) + real(2) * lg;
    trinv64_lds(Ls, Xs, rdiag);
    real* Xo = g.Linv + ((size_t)l * g.nblk + g.r0 / CB) * CB * CB;
    for (int e = threadIdx.x; e < CB * CB; e += blockDim.x) {
        const int i = e / CB, j = e % CB;
        Xo[e] = Xs[i][j];
        if (i < n && j < n) A[(size_t)i * g.lda + j] = Ls[i][j];
    }
}

// ---- 128-wide block steps ------------------------------------------------------------------------------------------
// A block step of the right-looking factorisation is a chain of dependent launches (diagonal block, panel solve, column
// update) with a 64 x 64 diagonal block on `batch` of 256 CUs at its head; at m = 800 that chain, not the flops, is the time.
// k_chol_diag2 takes a 128 x 128 diagonal block in ONE launch -- L11 = chol(A11), X11 = L11^-1, L21 = A21 X11^T,
// A22 -= L21 L21^T, L22 = chol(A22), X22 = L22^-1, X21 = -X22 (L21 X11), the 64^3 products on the f64 MFMA out of LDS --
// so the outer loop has half the steps, and its panel solves / trailing updates are K = 128 products (twice the arithmetic
// intensity of K = 64).  Writes L, both 64 x 64 inverse blocks (what trsm / potri consume) and the 128 x 128 inverse
// [[X11, 0], [X21, X22]] for the panel solve.
typedef double d4k __attribute__((ext_vector_type(4)));
// acc[u] (u = 0..3) = tile (ti, tj) = ((4 wave + u) / 4, (4 wave + u) % 4) of A (64 x 64) times B or B^T, operands in LDS
template <bool TB>
__device__ __forceinline__ void mm64_tiles(const real (*A)[CLD], const real (*B)[CLD], d4k* acc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = 4 * wave + u, ti = 16 * (t >> 2), tj = 16 * (t & 3);
        d4k c = {0, 0, 0, 0};
#pragma unroll
        for (int k0 = 0; k0 < CB; k0 += 4)
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ti + r][k0 + q], TB ? B[tj + r][k0 + q] : B[k0 + q][tj + r], c, 0, 0, 0);
        acc[u] = c;
    }
}
// element (row, col) of accumulator register e of tile u held by this lane
#define MM64_ROW(u, e) (16 * ((4 * (threadIdx.x >> 6) + (u)) >> 2) + ((threadIdx.x & 63) >> 4) + 4 * (e))
#define MM64_COL(u) (16 * ((4 * (threadIdx.x >> 6) + (u)) & 3) + (threadIdx.x & 15))

struct Diag2Args {
    int m, r0, n, lda, batch, first, nblk, nblk2;   // block rows r0 .. r0 + n (64 < n <= 128)
    long long sA;
    real* A;
    real* Linv;       // (batch, nblk, 64, 64)
    real* Linv2;      // (batch, nblk2, 128, 128): slot r0 / 128
    real* logdet;
    int onewave;
};
// Two 64 x 64 blocks of LDS (68 KB, as k_chol_diag), not four: a workgroup with 135 KB needs a CU with no GEMM workgroup on it
// (66.5 KB each, two per CU), and inside a training step the chip is full of them on the other streams and on the look-ahead
// branch -- the launch then waits for whole CUs to drain (265 us average in the SPRITES step against 100 us alone).  With 68 KB
// it fits beside one GEMM workgroup, i.e. into the first slot that retires.  A21, A22 and W = L21 X11 (stashed in the X21 slot
// of the 128 x 128 inverse, each thread re-reading the elements it wrote) stream through global memory instead.
#define DIAG2_LDS_BYTES ((2 * CB * CLD + 3 * CB) * sizeof(real))
__global__ __launch_bounds__(256) void k_chol_diag2(Diag2Args g) {
    extern __shared__ __align__(16) unsigned char diag_lds[];
    real (*colk)[CB] = reinterpret_cast<real (*)[CB]>(diag_lds);
    real* rdiag = reinterpret_cast<real*>(diag_lds + 2 * CB * sizeof(real));
    real (*P)[CLD] = reinterpret_cast<real (*)[CLD]>(diag_lds + 3 * CB * sizeof(real));
    real (*Q)[CLD] = reinterpret_cast<real (*)[CLD]>(diag_lds + (3 * CB + CB * CLD) * sizeof(real));
    const int l = blockIdx.x, n2 = g.n - CB;                 // rows of the second 64-block (1..64)
    real* A = g.A + (size_t)l * g.sA + (size_t)g.r0 * g.lda + g.r0;
    real* X1o = g.Linv + ((size_t)l * g.nblk + g.r0 / CB) * CB * CB;
    real* X2o = X1o + CB * CB;
    real* Xb = g.Linv2 + ((size_t)l * g.nblk2 + g.r0 / (2 * CB)) * (4 * CB * CB);        // 128 x 128, ld 128
    // ---- first 64-block: P = A11 -> L11, Q = X11
    for (int e = threadIdx.x; e < CB * CB; e += blockDim.x) P[e / CB][e % CB] = A[(size_t)(e / CB) * g.lda + e % CB];
    chol64(P, colk, rdiag, g.onewave);
    real lg = (threadIdx.x < CB) ? log(P[threadIdx.x][threadIdx.x]) : real(0);
    lg = wave_sum_c(lg);
    if (threadIdx.x == 0) g.logdet[l] = (g.first ? real(0) : g.logdet[l]) + real(2) * lg;
    trinv64_lds(P, Q, rdiag);
    for (int e = threadIdx.x; e < CB * CB; e += blockDim.x) {
        const int i = e / CB, j = e % CB;
        const real x = Q[i][j];
        X1o[e] = x;
        Xb[(size_t)i * 2 * CB + j] = x;
        Xb[(size_t)i * 2 * CB + CB + j] = real(0);
        A[(size_t)i * g.lda + j] = P[i][j];
    }
    __syncthreads();
    // ---- L21 = A21 X11^T (P = A21 -> L21)
    for (int e = threadIdx.x; e < CB * CB; e += blockDim.x) {
        const int i = e / CB, j = e % CB;
        P[i][j] = i < n2 ? A[(size_t)(CB + i) * g.lda + j] : real(0);
    }
    __syncthreads();
    d4k acc[4];
    mm64_tiles<true>(P, Q, acc);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) P[MM64_ROW(u, e)][MM64_COL(u)] = acc[u][e];
    __syncthreads();
    for (int e = threadIdx.x; e < CB * CB; e += blockDim.x) {
        const int i = e / CB, j = e % CB;
        if (i < n2) A[(size_t)(CB + i) * g.lda + j] = P[i][j];
    }
    // ---- W = L21 X11 -> the X21 slot of the 128 x 128 inverse (stash)
    mm64_tiles<false>(P, Q, acc);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) Xb[(size_t)(CB + MM64_ROW(u, e)) * 2 * CB + MM64_COL(u)] = acc[u][e];
    // ---- Q = A22 - L21 L21^T (identity pad; the pad rows of L21 are zero)
    mm64_tiles<true>(P, P, acc);
    __syncthreads();                                         // every read of X11 (Q) is done
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = MM64_ROW(u, e), j = MM64_COL(u);
            const real a22 = (i < n2 && j < n2) ? A[(size_t)(CB + i) * g.lda + CB + j] : (i == j ? real(1) : real(0));
            Q[i][j] = a22 - acc[u][e];
        }
    // ---- second 64-block: Q -> L22, P = X22
    chol64(Q, colk, rdiag, g.onewave);
    lg = (threadIdx.x < n2) ? log(Q[threadIdx.x][threadIdx.x]) : real(0);
    lg = wave_sum_c(lg);
    if (threadIdx.x == 0) g.logdet[l] += real(2) * lg;
    trinv64_lds(Q, P, rdiag);
    for (int e = threadIdx.x; e < CB * CB; e += blockDim.x) {
        const int i = e / CB, j = e % CB;
        const real x = P[i][j];
        X2o[e] = x;
        Xb[(size_t)(CB + i) * 2 * CB + CB + j] = x;
        if (i < n2 && j < n2) A[(size_t)(CB + i) * g.lda + CB + j] = Q[i][j];
    }
    __syncthreads();                                         // every read of L22 (Q) is done
    // ---- X21 = -X22 W (Q = W, each thread the elements it stashed)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) Q[MM64_ROW(u, e)][MM64_COL(u)] = Xb[(size_t)(CB + MM64_ROW(u, e)) * 2 * CB + MM64_COL(u)];
    __syncthreads();
    mm64_tiles<false>(P, Q, acc);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) Xb[(size_t)(CB + MM64_ROW(u, e)) * 2 * CB + MM64_COL(u)] = -acc[u][e];
}

// inverse of every 64 x 64 diagonal block of a lower-triangular L: grid (nblk, batch)
struct TriDiagArgs {
    int m, ldl, nblk;
    long long sL;
    const real* L;
    real* Linv;       // (batch, nblk, 64, 64)
};
__global__ __launch_bounds__(256) void k_tri_diag_inv(TriDiagArgs g) {
    extern __shared__ __align__(16) unsigned char diag_lds[];
    real (*Ls)[CLD] = reinterpret_cast<real (*)[CLD]>(diag_lds);
    real (*Xs)[CLD] = reinterpret_cast<real (*)[CLD]>(diag_lds + CB * CLD * sizeof(real));
    const int kb = blockIdx.x, l = blockIdx.y, r0 = kb * CB, n = min(CB, g.m - r0);
    const real* L = g.L + (size_t)l * g.sL + (size_t)r0 * g.ldl + r0;
    {   // 16 independent loads per thread in flight, then the LDS stores
        const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
        real r[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i = ti + 16 * a, j = tj + 16 * b;
                r[a][b] = (i < n && j < n && j <= i) ? L[(size_t)i * g.ldl + j] : (i == j ? real(1) : real(0));
            }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) Ls[ti + 16 * a][tj + 16 * b] = r[a][b];
    }
    real* rdiag = reinterpret_cast<real*>(diag_lds + (2 * CB * CLD + 2 * CB) * sizeof(real));
    __syncthreads();
    if (threadIdx.x < CB) rdiag[threadIdx.x] = real(1) / Ls[threadIdx.x][threadIdx.x];
    trinv64_lds(Ls, Xs, rdiag);
    real* Xo = g.Linv + ((size_t)l * g.nblk + kb) * CB * CB;
    for (int e = threadIdx.x; e < CB * CB; e += blockDim.x) Xo[e] = Xs[e / CB][e % CB];
}

// dst[l][r0 + i][c0 + j] = src[l][i][j] for an (nr x nc) block; grid (ceil(nr*nc/256), batch)
__global__ void k_copy_block(int nr, int nc, const real* __restrict__ src, int lds, long long ss, real* __restrict__ dst,
                             int ldd, long long sd) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)nr * nc) return;
    const int i = (int)(e / nc), j = (int)(e % nc), l = blockIdx.y;
    dst[(size_t)l * sd + (size_t)i * ldd + j] = src[(size_t)l * ss + (size_t)i * lds + j];
}
// strict upper triangle of every (m x m, ld) matrix <- 0
__global__ void k_zero_upper(int m, int ld, long long sA, real* __restrict__ A) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)m * m) return;
    const int i = (int)(e / m), j = (int)(e % m);
    if (j > i) A[(size_t)blockIdx.y * sA + (size_t)i * ld + j] = real(0);
}
// the ZBAND columns right of the diagonal of every (m x m, ld) matrix <- 0; grid (ceil(m * ZBAND / 256), batch)
__global__ void k_zero_band(int m, int ld, long long sA, real* __restrict__ A) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)m * 128) return;
    const int i = (int)(e / 128), j = i + 1 + (int)(e % 128);
    if (j < m) A[(size_t)blockIdx.y * sA + (size_t)i * ld + j] = real(0);
}
// diagonal blocks of X (m x m, zero above the diagonal where tiles read it) <- Linv blocks; grid (nblk * 16, batch)
__global__ void k_place_diag_blocks(int m, int nblk, const real* __restrict__ Linv, real* __restrict__ X) {
    const int kb = blockIdx.x / 16, l = blockIdx.y, r0 = kb * CB, n = min(CB, m - r0);
    const real* src = Linv + ((size_t)l * nblk + kb) * CB * CB;
    real* dst = X + (size_t)l * m * m + (size_t)r0 * m + r0;
    for (int e = (blockIdx.x % 16) * 256 + threadIdx.x; e < CB * CB; e += 16 * 256) {
        const int i = e / CB, j = e % CB;
        if (i < n && j < n) dst[(size_t)i * m + j] = src[e];
    }
}

// 128 x 128 diagonal blocks of X <- the inverse blocks k_chol_diag2 left in Linv2 (rows r0 .. r0 + n, 64 < n <= 128); grid (64, batch)
__global__ void k_place_diag_block2(int m, int r0, int n, int slot, int nblk2, const real* __restrict__ Linv2, real* __restrict__ X) {
    const int l = blockIdx.y;
    const real* src = Linv2 + ((size_t)l * nblk2 + slot) * (4 * CB * CB);
    real* dst = X + (size_t)l * m * m + (size_t)r0 * m + r0;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < 4 * CB * CB; e += 64 * 256) {
        const int i = e / (2 * CB), j = e % (2 * CB);
        if (i < n && j < n) dst[(size_t)i * m + j] = src[e];
    }
}
inline unsigned nblk256(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

#define RUNC(call) do { int rc__ = (call); if (rc__) return rc__; } while (0)

// ---- workspace sizes -----------------------------------------------------------------------------------------
extern "C" size_t svgp_potrf_workspace_elems(int m, int batch) {
    if (m < 1 || batch < 0) return 0;
    const size_t nblk = (size_t)(m + CB - 1) / CB;
    const size_t nblk2 = (size_t)(m + 2 * CB - 1) / (2 * CB);
    // L_kk^-1 blocks (64) + 128 x 128 inverses of the outer diagonal blocks + two scaled 128-wide panels (look-ahead)
    return (size_t)batch * (nblk * CB * CB + nblk2 * 4 * CB * CB + 2 * (size_t)m * 2 * CB);
}
extern "C" size_t svgp_trsm_workspace_elems(int m, int n, int batch) {
    if (m < 1 || n < 0 || batch < 0) return 0;
    const size_t nblk = (size_t)(m + CB - 1) / CB;
    return (size_t)batch * (nblk * CB * CB + (size_t)CB * n);            // L_kk^-1 blocks + one solved block
}
extern "C" size_t svgp_potri_workspace_elems(int m, int batch) {
    if (m < 1 || batch < 0) return 0;
    const size_t nblk = (size_t)(m + CB - 1) / CB, h = (size_t)(m + 1) / 2 + CB;
    return (size_t)batch * (nblk * CB * CB + (size_t)m * m + h * h);     // L_kk^-1 blocks + L^-1 + one product T
}

// ---- potrf ----------------------------------------------------------------------------------------------------
// A (batch, m, lda) SPD -> lower Cholesky factor in place (strict upper triangle zeroed), logdet[l] = log det A[l].
// On return work[0 .. batch * nblk * 4096) holds the inverses of the 64 x 64 diagonal blocks of L, (batch, nblk, 64, 64).
// band != 0 (svgp_potrf_batched_band, for callers inside the library that consume the factor through the tile GEMMs only, like
// svgp_spd_inverse_batched): instead of the whole strict upper triangle only the `ZBAND` columns right of the diagonal are
// zeroed -- what a tile of at most ZBAND x ZBAND that touches the diagonal can read of it.
#define ZBAND 128
static bool potrf_wide_steps(int m) {
    static const int wide_on = [] { const char* e = getenv("SVGP_POTRF_BLOCK"); return (e && atoi(e) == 64) ? 0 : 1; }();
    return wide_on && m >= 512;
}
static int potrf_impl(int m, int batch, double* A, int lda, long long strideA, double* logdet, double* work, void* stream, int band) {
    SVGP_REQUIRE(m >= 1 && batch >= 0 && lda >= m, SVGP_ERR_INVALID, "bad m / batch / lda (m=%d batch=%d lda=%d)", m, batch, lda);
    if (batch == 0) return SVGP_OK;
    SVGP_REQUIRE(A && logdet && work, SVGP_ERR_INVALID, "NULL device pointer");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (m + CB - 1) / CB, nblk2 = (m + 2 * CB - 1) / (2 * CB);
    real* Linv = work;
    real* Linv2 = work + (size_t)batch * nblk * CB * CB;        // (batch, nblk2, 128, 128): inverses of the outer diagonal blocks
    real* Pn2 = Linv2 + (size_t)batch * nblk2 * 4 * CB * CB;    // 2 x (batch, m, W): scaled panels of the current / previous step
    SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_diag), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)DIAG_LDS_BYTES));
    SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_diag2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)DIAG2_LDS_BYTES));
    DiagArgs d;
    d.m = m; d.lda = lda; d.batch = batch; d.sA = strideA; d.A = A; d.Linv = Linv; d.nblk = nblk; d.logdet = logdet;
    Diag2Args d2;
    d2.m = m; d2.lda = lda; d2.batch = batch; d2.sA = strideA; d2.A = A; d2.Linv = Linv; d2.Linv2 = Linv2; d2.nblk = nblk;
    d2.nblk2 = nblk2; d2.logdet = logdet;
    // diagonal blocks on one wave with a row per lane (chol64_wave); SVGP_CHOL_WAVE=0: the four-wave register-tiled form
    static const int onewave = [] { const char* e = getenv("SVGP_CHOL_WAVE"); return (e && e[0] == '0') ? 0 : 1; }();
    d.onewave = d2.onewave = onewave;
    // Block steps of W = 128 rows from m >= 512 (k_chol_diag2: half the dependent launches, K = 128 products; the triangular inverse
    // of svgp_spd_inverse_batched then starts from 128-blocks: 512 x 16 inverse 557 -> 522 us), 64 below
    // (SVGP_POTRF_BLOCK=64 forces the 64-wide steps).
    const int W = potrf_wide_steps(m) ? 2 * CB : CB, nstep = (m + W - 1) / W;   // (256 x 17: 243 -> 255 us, 512 x 16: 553 -> 561 with the wide steps)
    // Look-ahead (right-looking with the trailing update split): after panel k is solved, ONLY block column k + 1 is updated on
    // the caller's stream -- that is all the next diagonal block and the next panel solve need -- and the rest of the trailing
    // update (columns k + 2 ..) runs on a side branch beside them: a block step costs max(trailing update, column update +
    // diagonal block + panel solve) instead of their sum.  SVGP_POTRF_LOOKAHEAD=0: everything in line.  Same operations on the
    // same values.
    static const int la_on = [] { const char* e = getenv("SVGP_POTRF_LOOKAHEAD"); return (e && e[0] == '0') ? 0 : 1; }();
    const bool la = la_on && m >= 640;            // (below, the fork / join per block step costs more than it hides: 512 x 16 553 -> 578 us)
    void* side = stream;
    bool side_open = false;
    struct JoinGuard {           // an error return inside the loop must not leave the look-ahead branch forked
        void* st; bool* open;
        ~JoinGuard() { if (*open) (void)svgp_side_branch_join(st, 0); }
    } join_guard{stream, &side_open};
    for (int kb = 0; kb < nstep; ++kb) {
        const int r0 = kb * W, nbk = m - r0 < W ? m - r0 : W, rem = m - r0 - nbk;
        real* Pn = Pn2 + (size_t)(kb & 1) * batch * m * W;
        const real* Lk;             // inverse of the diagonal block's factor, row-major, leading dimension ldk
        int ldk;
        long long sLk;
        if (nbk > CB) {
            d2.r0 = r0; d2.n = nbk; d2.first = kb == 0;
            hipLaunchKernelGGL(k_chol_diag2, dim3(batch), dim3(256), DIAG2_LDS_BYTES, s, d2);
            Lk = Linv2 + (size_t)kb * 4 * CB * CB; ldk = 2 * CB; sLk = (long long)nblk2 * 4 * CB * CB;
        } else {
            d.r0 = r0; d.nbk = nbk; d.first = kb == 0;
            hipLaunchKernelGGL(k_chol_diag, dim3(batch), dim3(256), DIAG_LDS_BYTES, s, d);
            Lk = Linv + (size_t)(r0 / CB) * CB * CB; ldk = CB; sLk = (long long)nblk * CB * CB;
        }
        SVGP_LAUNCH_CHECK();
        if (rem == 0) break;
        real* panel = A + (size_t)(r0 + nbk) * lda + r0;        // A[r0+nbk:, r0:r0+nbk]
        real* trail = A + (size_t)(r0 + nbk) * lda + r0 + nbk;
        // L_ik = A_ik L_kk^-T   (the inverse stored row-major: B^T form) into the contiguous panel copy Pn the updates read, then
        // back into the matrix.  (Not as a second output of the product: with the 64-wide tiles this short contraction takes, two
        // tile columns span the panel and one would overwrite operand columns the other is still reading.)
        RUNC(svgp_dgemm_tri_batched(0, 0, 1, rem, nbk, nbk, 1.0, panel, lda, strideA, Lk, ldk, sLk, 0.0, Pn, W, (long long)m * W,
                                    batch, stream));
        hipLaunchKernelGGL(k_copy_block, dim3(nblk256((long long)rem * nbk), batch), dim3(256), 0, s, rem, nbk, Pn, W,
                           (long long)m * W, panel, lda, strideA);
        SVGP_LAUNCH_CHECK();
        const int nb1 = rem < W ? rem : W, rem2 = rem - nb1;     // next block column / what lies beyond it
        if (!la || rem2 == 0) {
            if (side_open) { side_open = false; RUNC(svgp_side_branch_join(stream, 0)); }
            // A_ij -= L_ik L_jk^T on the tiles that touch the lower triangle
            RUNC(svgp_dgemm_tri_batched(1, 0, 1, rem, rem, nbk, -1.0, Pn, W, (long long)m * W, Pn, W, (long long)m * W, 1.0,
                                        trail, lda, strideA, batch, stream));
            continue;
        }
        // the previous step's side update wrote column k + 1 too: it must be complete before this step's column update
        if (side_open) { side_open = false; RUNC(svgp_side_branch_join(stream, 0)); }
        // column k + 1 (rows r1 .., nb1 columns): A[r1:, r1:r1+nb1] -= Pn[r1-rows] Pn[block k + 1 rows]^T
        RUNC(svgp_dgemm_tri_batched(1, 0, 1, rem, nb1, nbk, -1.0, Pn, W, (long long)m * W, Pn, W, (long long)m * W, 1.0,
                                    trail, lda, strideA, batch, stream));
        // columns k + 2 ..: on the side branch (forked here: after the panel solve and, through stream order, after the join)
        RUNC(svgp_side_branch_fork(stream, &side, 0));
        side_open = true;
        RUNC(svgp_dgemm_tri_batched(1, 0, 1, rem2, rem2, nbk, -1.0, Pn + (size_t)nb1 * W, W, (long long)m * W,
                                    Pn + (size_t)nb1 * W, W, (long long)m * W, 1.0, trail + (size_t)nb1 * lda + nb1, lda, strideA,
                                    batch, side));
    }
    if (side_open) { side_open = false; RUNC(svgp_side_branch_join(stream, 0)); }
    if (band) hipLaunchKernelGGL(k_zero_band, dim3(nblk256((long long)m * ZBAND), batch), dim3(256), 0, s, m, lda, strideA, A);
    else hipLaunchKernelGGL(k_zero_upper, dim3(nblk256((long long)m * m), batch), dim3(256), 0, s, m, lda, strideA, A);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_potrf_batched(int m, int batch, double* A, int lda, long long strideA, double* logdet, double* work,
                                  void* stream) {
    return potrf_impl(m, batch, A, lda, strideA, logdet, work, stream, 0);
}
int svgp_potrf_batched_band(int m, int batch, double* A, int lda, long long strideA, double* logdet, double* work, void* stream) {
    return potrf_impl(m, batch, A, lda, strideA, logdet, work, stream, 1);
}

// ---- trsm -----------------------------------------------------------------------------------------------------
// side 0: op(L) X = B, B (batch, m, n);  side 1: X op(L) = B, B (batch, n, m);  trans: op(L) = L^T.  L (m x m, ldl)
// lower triangular (strict upper part ignored), strideL 0 = one L for the whole batch.  B is overwritten by X.
extern "C" int svgp_trsm_batched(int side, int trans, int m, int n, const double* L, int ldl, long long strideL, double* B,
                                 int ldb, long long strideB, int batch, double* work, void* stream) {
    SVGP_REQUIRE(m >= 1 && n >= 0 && batch >= 0 && ldl >= m, SVGP_ERR_INVALID, "bad shape m=%d n=%d batch=%d ldl=%d", m, n, batch, ldl);
    SVGP_REQUIRE((side == 0 || side == 1) && (trans == 0 || trans == 1), SVGP_ERR_INVALID, "side / trans must be 0 or 1");
    SVGP_REQUIRE(ldb >= (side == 0 ? n : m), SVGP_ERR_INVALID, "ldb=%d too small", ldb);
    if (batch == 0 || n == 0) return SVGP_OK;
    SVGP_REQUIRE(L && B && work, SVGP_ERR_INVALID, "NULL device pointer");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (m + CB - 1) / CB, nL = strideL == 0 ? 1 : batch;
    real* Linv = work;
    real* T = work + (size_t)batch * nblk * CB * CB;            // (batch, 64, n) or (batch, n, 64): the solved block
    const long long sLi = strideL == 0 ? 0 : (long long)nblk * CB * CB;
    TriDiagArgs td;
    td.m = m; td.ldl = ldl; td.nblk = nblk; td.sL = strideL; td.L = L; td.Linv = Linv;
    SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tri_diag_inv), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)DIAG_LDS_BYTES));
    hipLaunchKernelGGL(k_tri_diag_inv, dim3(nblk, nL), dim3(256), DIAG_LDS_BYTES, s, td);
    SVGP_LAUNCH_CHECK();
    // forward over the blocks when (left, no-trans) or (right, trans); backward otherwise
    const bool fwd = (side == 0) == (trans == 0);
    for (int step = 0; step < nblk; ++step) {
        const int kb = fwd ? step : nblk - 1 - step, r0 = kb * CB, nbk = m - r0 < CB ? m - r0 : CB;
        const real* Lk = Linv + (size_t)kb * CB * CB;
        const int lo = fwd ? r0 + nbk : 0, cnt = fwd ? m - r0 - nbk : r0;        // the blocks still to be updated
        if (side == 0) {
            real* Bk = B + (size_t)r0 * ldb;
            // X_k = op(Linv_kk) B_k
            RUNC(svgp_dgemm_batched(trans, 0, nbk, n, nbk, 1.0, Lk, CB, sLi, Bk, ldb, strideB, 0.0, T, n, (long long)CB * n,
                                    batch, stream));
            hipLaunchKernelGGL(k_copy_block, dim3(nblk256((long long)nbk * n), batch), dim3(256), 0, s, nbk, n, T, n,
                               (long long)CB * n, Bk, ldb, strideB);
            SVGP_LAUNCH_CHECK();
            if (cnt > 0) {
                // no-trans: B_i -= L[i, k] X_k (i > k);  trans: B_i -= L[k, i]^T X_k (i < k)
                const real* Lik = trans ? L + (size_t)r0 * ldl + lo : L + (size_t)lo * ldl + r0;
                RUNC(svgp_dgemm_batched(trans, 0, cnt, n, nbk, -1.0, Lik, ldl, strideL, T, n, (long long)CB * n, 1.0,
                                        B + (size_t)lo * ldb, ldb, strideB, batch, stream));
            }
        } else {
            real* Bk = B + r0;                                   // columns r0 .. r0 + nbk of the (n x m) right-hand side
            // X_k = B_k op(Linv_kk)
            RUNC(svgp_dgemm_batched(0, trans, n, nbk, nbk, 1.0, Bk, ldb, strideB, Lk, CB, sLi, 0.0, T, CB, (long long)n * CB,
                                    batch, stream));
            hipLaunchKernelGGL(k_copy_block, dim3(nblk256((long long)n * nbk), batch), dim3(256), 0, s, n, nbk, T, CB,
                               (long long)n * CB, Bk, ldb, strideB);
            SVGP_LAUNCH_CHECK();
            if (cnt > 0) {
                // no-trans (backward): B[:, j] -= X_k L[k, j] (j < k);  trans (forward): B[:, j] -= X_k L[j, k]^T (j > k)
                const real* Lkj = trans ? L + (size_t)lo * ldl + r0 : L + (size_t)r0 * ldl + lo;
                RUNC(svgp_dgemm_batched(0, trans, n, cnt, nbk, -1.0, T, CB, (long long)n * CB, Lkj, ldl, strideL, 1.0,
                                        B + lo, ldb, strideB, batch, stream));
            }
        }
    }
    return SVGP_OK;
}

// ---- potri ----------------------------------------------------------------------------------------------------
namespace {
// X[lo:hi, lo:hi] (in 64-blocks) <- inverse of the lower-triangular L[lo:hi, lo:hi]; diagonal blocks are already placed.
int trtri_rec(int m, int batch, const real* L, real* X, real* T, int blo, int bhi, void* stream, int bw = CB) {
    if (bhi - blo <= 1) return SVGP_OK;
    const int bmid = blo + (bhi - blo + 1) / 2;
    RUNC(trtri_rec(m, batch, L, X, T, blo, bmid, stream, bw));
    RUNC(trtri_rec(m, batch, L, X, T, bmid, bhi, stream, bw));
    const int r1 = blo * bw, r2 = bmid * bw, r3 = bhi * bw < m ? bhi * bw : m, n1 = r2 - r1, n2 = r3 - r2;
    const long long mm = (long long)m * m, sT = (long long)n2 * n1;
    // T = L21 X11  (X11 lower triangular: the contraction starts at the tile's first column)
    RUNC(svgp_dgemm_tri_batched(4, 0, 0, n2, n1, n1, 1.0, L + (size_t)r2 * m + r1, m, mm, X + (size_t)r1 * m + r1, m, mm, 0.0,
                                T, n1, sT, batch, stream));
    // X21 = -X22 T  (X22 lower triangular: the contraction ends with the tile's last row)
    RUNC(svgp_dgemm_tri_batched(8, 0, 0, n2, n1, n2, -1.0, X + (size_t)r2 * m + r2, m, mm, T, n1, sT, 0.0,
                                X + (size_t)r2 * m + r1, m, mm, batch, stream));
    return SVGP_OK;
}
}  // namespace

// A (batch, m, m) contiguous holding the factor L of svgp_potrf_batched -> A^-1 = L^-T L^-1 (full symmetric matrix).
// `linv_blocks` = the potrf workspace head (inverses of the diagonal blocks of L), or NULL: they are recomputed.
// linv2 != NULL (svgp_potri_batched_wide, after a factorisation that ran with 128-wide block steps): the 128 x 128 inverse
// blocks of those steps are placed as they are and the recursion starts one level higher -- half the leaves, and the twelve
// (m = 800) smallest product launches of the halving, which are launch-bound, disappear.
static int potri_impl(int m, int batch, double* A, const double* linv_blocks, const double* linv2, double* work, void* stream) {
    SVGP_REQUIRE(m >= 1 && batch >= 0, SVGP_ERR_INVALID, "bad m / batch");
    if (batch == 0) return SVGP_OK;
    SVGP_REQUIRE(A && work, SVGP_ERR_INVALID, "NULL device pointer");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (m + CB - 1) / CB;
    const long long mm = (long long)m * m;
    real* Linv = work;
    real* X = work + (size_t)batch * nblk * CB * CB;
    real* T = X + (size_t)batch * mm;
    if (linv_blocks == nullptr) {
        TriDiagArgs td;
        td.m = m; td.ldl = m; td.nblk = nblk; td.sL = mm; td.L = A; td.Linv = Linv;
        SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tri_diag_inv),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)DIAG_LDS_BYTES));
        hipLaunchKernelGGL(k_tri_diag_inv, dim3(nblk, batch), dim3(256), DIAG_LDS_BYTES, s, td);
        SVGP_LAUNCH_CHECK();
        linv_blocks = Linv;
    }
    // X above the diagonal: only the band a diagonal-touching tile reads (the rest of the upper triangle is never read: every
    // product below runs with contraction ranges cut to the tile's rows / columns)
    hipLaunchKernelGGL(k_zero_band, dim3(nblk256((long long)m * 128), batch), dim3(256), 0, s, m, m, (long long)mm, X);
    SVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_place_diag_blocks, dim3(nblk * 16, batch), dim3(256), 0, s, m, nblk, linv_blocks, X);
    SVGP_LAUNCH_CHECK();
    if (linv2) {
        const int nblk2 = (m + 2 * CB - 1) / (2 * CB);
        for (int kb = 0; kb < nblk2; ++kb) {
            const int r0 = kb * 2 * CB, n = m - r0 < 2 * CB ? m - r0 : 2 * CB;
            if (n <= CB) continue;                      // a last step of at most 64 rows was a plain 64-block
            hipLaunchKernelGGL(k_place_diag_block2, dim3(64, batch), dim3(256), 0, s, m, r0, n, kb, nblk2, linv2, X);
            SVGP_LAUNCH_CHECK();
        }
        RUNC(trtri_rec(m, batch, A, X, T, 0, nblk2, stream, 2 * CB));
    } else {
        RUNC(trtri_rec(m, batch, A, X, T, 0, nblk, stream));
    }
    // A^-1 = X^T X, X lower triangular: the contraction starts at max(first row, first column) of the tile; the result is
    // symmetric, so only the tiles that touch the lower triangle are computed and stored mirrored (as LAPACK's potri returns one
    // triangle): 28 instead of 49 tiles per matrix at m = 800 (SVGP_POTRI_FULL=1: both triangles computed)
    static const int potri_full = [] { const char* e = getenv("SVGP_POTRI_FULL"); return (e && e[0] == '1') ? 1 : 0; }();
    RUNC(svgp_dgemm_tri_batched(potri_full ? (2 | 4) : (1 | 2 | 4 | 16), 1, 0, m, m, m, 1.0, X, m, mm, X, m, mm, 0.0, A, m, mm, batch,
                                stream));
    return SVGP_OK;
}
extern "C" int svgp_potri_batched(int m, int batch, double* A, const double* linv_blocks, double* work, void* stream) {
    return potri_impl(m, batch, A, linv_blocks, nullptr, work, stream);
}
// `potrf_work` = the workspace svgp_potrf_batched[_band] has just filled for the same (m, batch)
int svgp_potri_batched_wide(int m, int batch, double* A, const double* potrf_work, double* work, void* stream) {
    const int nblk = (m + CB - 1) / CB;
    const double* linv2 = potrf_wide_steps(m) ? potrf_work + (size_t)batch * nblk * CB * CB : nullptr;
    return potri_impl(m, batch, A, potrf_work, linv2, work, stream);
}
