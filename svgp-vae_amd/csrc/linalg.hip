// Batched float64 linear algebra on global-memory matrices for the large-m (m > 64) GP path:
//   svgp_dgemm_batched        C[l] = alpha op(A[l]) op(B[l]) + beta C[l]     (f64 MFMA, 128x128 / 64x64 tiles)
//   svgp_spd_inverse_batched  A[l] <- A[l]^-1, logdet[l]                      (blocked Gauss-Jordan)
// They replace tf.matmul / tf.linalg.inv / tf.linalg.cholesky+log(diag) of the reference
// (SVGPVAE_model.py:239,270-274,319,328-341) when the m x m matrices no longer fit in LDS.
#include "common.hpp"
#include "sweep32.hpp"
#include <cstdlib>
#include <type_traits>

namespace {

typedef double d4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// GEMM.  Workgroup = 256 threads = 4 waves; output tile 32 WT x 32 WT; wave w owns the quadrant
// (w>>1, w&1) = WT x WT MFMA 16x16 tiles; k-panels of 16 staged in LDS as As[k][i], Bs[k][j] (k-major:
// the MFMA operand fetch A[i=lane&15][k=lane>>4] walks 16 consecutive i -> conflict-free).
// ---------------------------------------------------------------------------------------------
// panel depth: 16 for the 128-tiles (64 MFMAs per wave per panel already cover the load latency), 32 for the 64- and
// 32-tiles (16 / 4 MFMAs per panel of 16 do not when one workgroup per CU is all the grid offers: config-3 statistics)
#define GK_OF(WT) 16

struct GemmArgs {
    int M, N, K;            // C is M x N, contraction K
    int tiles_m, tiles_n, batch, xcd_remap;
    int full_m, full_n;     // k_gemm_mixed: tile rows / columns [0, full) are 128 wide, the last one (if any) is an edge tile
    int ta, tb;             // op(A) = A^T if ta (A stored K x M), op(B) = B^T if tb (B stored N x K)
    int lda, ldb, ldc;
    long long sa, sb, sc;   // batch strides in elements (0 = shared)
    real alpha, beta;
    const void* A; const void* B; void* C;      // element type = the kernel's storage type TS
    // triangular structure (svgp_dgemm_tri_batched): bit 0 = only tiles that touch the lower triangle (j0 <= i0 + tile - 1);
    // bit 1 / bit 2 = contraction starts at the tile's first row / first column (operands that are zero for k < i / k < j);
    // bit 3 = contraction ends with the tile's last row (operand zero for k > i);
    // bit 4 (with bit 0, M == N) = the result is symmetric: every computed tile below the diagonal is also stored transposed
    int tri;
    // optional weights of the contraction index: op(B)[k][j] is multiplied by wk[k * ldw + l * sw] while it is staged
    // (S_l = Kn^T diag(w_l) Kn without materialising diag(w_l) Kn); element type = TS
    const void* wk; int ldw; long long sw;
    // optional transform of the B operand while it is staged: op(B) := Bsub - op(B), Bsub stored like B (leading dimension ldb),
    // shared by the batch (H = G (Ki - Aji) without materialising Ki - Aji: one pass over an (L, m, m) array less)
    const void* bsub;
    // extended epilogue (svgp_gemm_epi, common.hpp; float64 storage): out1 = alpha acc + beta C + g1 E + d1 I -> C,
    // out2 = a2 acc + g2 E + d2 I -> C2 (same leading dimension as C); E: lde, batch stride se (0 = shared)
    int epi_on;
    const void* E; int lde; long long se; void* C2; long long sc2; int ldc2;
    real g1, d1, a2, g2, d2;
    const real* alpha_dev;  // extended epilogue only: alpha *= *alpha_dev (device scalar), NULL = 1
    int e_sym;              // E exactly symmetric: the mirrored store reuses E[i][j]
};

// k-panels of 16 are double-buffered in LDS and a third one is in flight in registers (see the main loop), one
// barrier per panel.  Loads are arranged so that 16 lanes cover 128 (f64) / 64 (f32) contiguous
// bytes of the operand whichever way it is stored.  WT = 4: 128 x 128 tile, 16 flop per staged byte, used
// when that still gives >= 192 workgroups; WT = 2: 64 x 64 tile for small problems.
// TS = storage type of A, B, C; TC = arithmetic type of the MFMA (operands converted while staging, accumulation in TC):
//   <double, double>  the float64 GEMM                       v_mfma_f64_16x16x4_f64
//   <double, float>   float64 matrices, float32 arithmetic   v_mfma_f32_16x16x4_f32 (twice the matrix rate, half the LDS)
//   <float,  float>   the float32 GEMM
typedef float f4_t __attribute__((ext_vector_type(4)));
#ifndef GEMM_PIPE
#define GEMM_PIPE 1
#endif
template <typename TC> struct MfmaT;
template <> struct MfmaT<double> {
    typedef d4_t acc_t;
    static constexpr int PAD = 2;          // LDS row pad (elements): conflict-free 16-wide operand fetch
    __device__ static __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int q, int e) { return q + 4 * e; }       // C/D: row = (lane >> 4) + 4 reg
};
template <> struct MfmaT<float> {
    typedef f4_t acc_t;
    static constexpr int PAD = 16;         // q rows 16 banks apart: the 32-lane ds_read_b32 group covers 32 banks
    __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int q, int e) { return 4 * q + e; }       // C/D: row = 4 (lane >> 4) + reg
};

// One output tile of 32 WM x 32 WN (wave (w >> 1, w & 1) owns WM x WN MFMA tiles) at rows i0.., columns j0.. of matrix l.
template <bool TA, bool TB, int WM, int WN, typename TS, typename TC, bool VEC>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, int l, int i0, int j0, TC* hs) {
    typedef MfmaT<TC> MF;
    typedef TS TS2 __attribute__((ext_vector_type(2)));
    typedef TC TC2 __attribute__((ext_vector_type(2)));
    constexpr int GK = 16, HM = 32 * WM, HN = 32 * WN, LDA_ = HM + MF::PAD, LDB_ = HN + MF::PAD;
    constexpr int NPA = HM * GK / 512, NPB = HN * GK / 512;
    static_assert(LDA_ % 2 == 0 && LDB_ % 2 == 0, "staging map below: 8 k-pairs x 32 rows per 256 threads");
    TC* As = hs;                         // [2][GK][LDA_]
    TC* Bs = hs + 2 * GK * LDA_;         // [2][GK][LDB_]
    if ((g.tri & 1) && j0 > i0 + HM - 1) return;
    int klo = 0, khi = g.K;
    if (g.tri & 2) klo = i0;
    if ((g.tri & 4) && j0 > klo) klo = j0;
    if ((g.tri & 8) && i0 + HM < khi) khi = i0 + HM;
    const TS* __restrict__ A = static_cast<const TS*>(g.A) + (size_t)l * g.sa;
    const TS* __restrict__ B = static_cast<const TS*>(g.B) + (size_t)l * g.sb;
    TS* C = static_cast<TS*>(g.C) + (size_t)l * g.sc;
    const TS* __restrict__ wk = g.wk ? static_cast<const TS*>(g.wk) + (size_t)l * g.sw : nullptr;
    const TS* __restrict__ Bs2 = static_cast<const TS*>(g.bsub);          // (never together with wk: rw holds either)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wi = (wave >> 1) * 16 * WM, wj = (wave & 1) * 16 * WN;
    typename MF::acc_t acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) acc[a][b] = typename MF::acc_t{0, 0, 0, 0};
    // Staging: NPA / NPB pairs of memory-adjacent elements per operand per thread.
    //   operand stored [x][k] (k contiguous): the pair (k, k + 1), k = 2 (tid & 7), of row x = (tid >> 3) + 32 h
    //   operand stored [k][x] (x contiguous): pair e = tid + 256 h of the GK x H / 2 panel, k = e / (H / 2), x = 2 (e % (H / 2))
    // A tile that lies inside the matrix takes its full panels without bounds checks, as one 16-byte load per pair when
    // the host found the operands 16-byte aligned with even leading dimensions (VEC): 60 -> 64 TFLOP/s on exact tiles
    // (tools/micro/gemm_stages.hip), and the checks were worth another 5 %.
    // A tile whose rows / columns stick out of the matrix still takes the unchecked path: the out-of-range row / column indices are
    // CLAMPED to the last valid one (their products land in accumulator rows that are never stored).  Only an operand whose
    // memory-adjacent pair runs along the clamped index needs that extent to be even.
    const bool a_ok = i0 + HM <= g.M || !TA || g.M % 2 == 0;
    const bool b_ok = j0 + HN <= g.N || TB || g.N % 2 == 0;
    const bool tile_fast = a_ok && b_ok && g.M >= 2 && g.N >= 2;
    TC ra[2 * NPA], rb[2 * NPB];
    TC rw[2 * NPB];        // contraction weights of the fetched B pairs: loaded with them, applied when the panel is staged (a product
                           // right behind the load would park the wave until the load returns)
    // instructions of the steady-state pipeline that go behind each of the first / last eight MFMAs of group 0
    constexpr int NDSW = (TA ? NPA : 2 * NPA) + (TB ? 2 * NPB : NPB), NMF = WM * WN;
    constexpr int PIPE_DSW = (NDSW + NMF / 2 - 1) / (NMF / 2 > 0 ? NMF / 2 : 1), PIPE_VMEM = (NPA + NPB + NMF / 2 - 1) / (NMF / 2 > 0 ? NMF / 2 : 1),
                  PIPE_VALU = 3;
    auto ldfast = [&](const TS* __restrict__ p, TC& v0, TC& v1) {
        if (VEC) {
            const TS2 v = *reinterpret_cast<const TS2*>(p);
            v0 = (TC)v.x; v1 = (TC)v.y;
        } else {
            v0 = (TC)p[0]; v1 = (TC)p[1];
        }
    };
    auto ldchk = [&](const TS* __restrict__ p, bool ok0, bool ok1, TC& v0, TC& v1) {
        v0 = ok0 ? (TC)p[0] : TC(0);
        v1 = ok1 ? (TC)p[1] : TC(0);
    };
    // one uniform branch per panel: full panel of a clampable tile -> straight-line unchecked loads, else the checked form
    // (the per-pair branches of a mixed form were ~100 instructions per panel in front of MFMA group 1)
    // straight-line unchecked loads of a full panel (tile_fast)
    auto fetch_fast = [&](int k0, auto hw) {
#pragma unroll
        for (int h = 0; h < NPA; ++h) {
            const int e = tid + 256 * h, kx = e / (HM / 2), xx = 2 * (e % (HM / 2));    // [k][x] map
            const int kk = 2 * (tid & 7), xk = (tid >> 3) + 32 * h;                      // [x][k] map
            if (TA) ldfast(A + (size_t)(k0 + kx) * g.lda + min(i0 + xx, g.M - 2), ra[2 * h], ra[2 * h + 1]);
            else ldfast(A + (size_t)min(i0 + xk, g.M - 1) * g.lda + k0 + kk, ra[2 * h], ra[2 * h + 1]);
        }
#pragma unroll
        for (int h = 0; h < NPB; ++h) {
            const int e = tid + 256 * h, kx = e / (HN / 2), xx = 2 * (e % (HN / 2));
            const int kk = 2 * (tid & 7), xk = (tid >> 3) + 32 * h;
            if (TB) ldfast(B + (size_t)min(j0 + xk, g.N - 1) * g.ldb + k0 + kk, rb[2 * h], rb[2 * h + 1]);
            else ldfast(B + (size_t)(k0 + kx) * g.ldb + min(j0 + xx, g.N - 2), rb[2 * h], rb[2 * h + 1]);
            if constexpr (decltype(hw)::value == 1) {
                if (TB) { rw[2 * h] = (TC)wk[(size_t)(k0 + kk) * g.ldw]; rw[2 * h + 1] = (TC)wk[(size_t)(k0 + kk + 1) * g.ldw]; }
                else rw[2 * h] = rw[2 * h + 1] = (TC)wk[(size_t)(k0 + kx) * g.ldw];
            }
            if constexpr (decltype(hw)::value == 2) {
                if (TB) ldfast(Bs2 + (size_t)min(j0 + xk, g.N - 1) * g.ldb + k0 + kk, rw[2 * h], rw[2 * h + 1]);
                else ldfast(Bs2 + (size_t)(k0 + kx) * g.ldb + min(j0 + xx, g.N - 2), rw[2 * h], rw[2 * h + 1]);
            }
        }
    };
    auto fetch = [&](int k0) {
        if (tile_fast && k0 + GK <= g.K) {
#pragma unroll
            for (int h = 0; h < NPA; ++h) {
                const int e = tid + 256 * h, kx = e / (HM / 2), xx = 2 * (e % (HM / 2));    // [k][x] map
                const int kk = 2 * (tid & 7), xk = (tid >> 3) + 32 * h;                      // [x][k] map
                if (TA) ldfast(A + (size_t)(k0 + kx) * g.lda + min(i0 + xx, g.M - 2), ra[2 * h], ra[2 * h + 1]);
                else ldfast(A + (size_t)min(i0 + xk, g.M - 1) * g.lda + k0 + kk, ra[2 * h], ra[2 * h + 1]);
            }
#pragma unroll
            for (int h = 0; h < NPB; ++h) {
                const int e = tid + 256 * h, kx = e / (HN / 2), xx = 2 * (e % (HN / 2));
                const int kk = 2 * (tid & 7), xk = (tid >> 3) + 32 * h;
                if (TB) {
                    const int gk = k0 + kk;
                    ldfast(B + (size_t)min(j0 + xk, g.N - 1) * g.ldb + gk, rb[2 * h], rb[2 * h + 1]);
                    if (wk) { rw[2 * h] = (TC)wk[(size_t)gk * g.ldw]; rw[2 * h + 1] = (TC)wk[(size_t)(gk + 1) * g.ldw]; }
                    if (Bs2) ldfast(Bs2 + (size_t)min(j0 + xk, g.N - 1) * g.ldb + gk, rw[2 * h], rw[2 * h + 1]);
                } else {
                    const int gk = k0 + kx;
                    ldfast(B + (size_t)gk * g.ldb + min(j0 + xx, g.N - 2), rb[2 * h], rb[2 * h + 1]);
                    if (wk) rw[2 * h] = rw[2 * h + 1] = (TC)wk[(size_t)gk * g.ldw];
                    if (Bs2) ldfast(Bs2 + (size_t)gk * g.ldb + min(j0 + xx, g.N - 2), rw[2 * h], rw[2 * h + 1]);
                }
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < NPA; ++h) {
            const int e = tid + 256 * h, kx = e / (HM / 2), xx = 2 * (e % (HM / 2));    // [k][x] map
            const int kk = 2 * (tid & 7), xk = (tid >> 3) + 32 * h;                      // [x][k] map
            if (TA) {
                const int gi = i0 + xx, gk = k0 + kx;
                ldchk(A + (size_t)gk * g.lda + gi, gk < g.K && gi < g.M, gk < g.K && gi + 1 < g.M, ra[2 * h], ra[2 * h + 1]);
            } else {
                const int gi = i0 + xk, gk = k0 + kk;
                ldchk(A + (size_t)gi * g.lda + gk, gi < g.M && gk < g.K, gi < g.M && gk + 1 < g.K, ra[2 * h], ra[2 * h + 1]);
            }
        }
#pragma unroll
        for (int h = 0; h < NPB; ++h) {
            const int e = tid + 256 * h, kx = e / (HN / 2), xx = 2 * (e % (HN / 2));
            const int kk = 2 * (tid & 7), xk = (tid >> 3) + 32 * h;
            if (TB) {
                const int gj = j0 + xk, gk = k0 + kk;
                ldchk(B + (size_t)gj * g.ldb + gk, gj < g.N && gk < g.K, gj < g.N && gk + 1 < g.K, rb[2 * h], rb[2 * h + 1]);
                if (wk) {
                    rw[2 * h] = gk < g.K ? (TC)wk[(size_t)gk * g.ldw] : TC(0);
                    rw[2 * h + 1] = gk + 1 < g.K ? (TC)wk[(size_t)(gk + 1) * g.ldw] : TC(0);
                }
                if (Bs2) ldchk(Bs2 + (size_t)gj * g.ldb + gk, gj < g.N && gk < g.K, gj < g.N && gk + 1 < g.K, rw[2 * h], rw[2 * h + 1]);
            } else {
                const int gj = j0 + xx, gk = k0 + kx;
                ldchk(B + (size_t)gk * g.ldb + gj, gk < g.K && gj < g.N, gk < g.K && gj + 1 < g.N, rb[2 * h], rb[2 * h + 1]);
                if (wk) rw[2 * h] = rw[2 * h + 1] = gk < g.K ? (TC)wk[(size_t)gk * g.ldw] : TC(0);
                if (Bs2) ldchk(Bs2 + (size_t)gk * g.ldb + gj, gk < g.K && gj < g.N, gk < g.K && gj + 1 < g.N, rw[2 * h], rw[2 * h + 1]);
            }
        }
    };
    auto stage = [&](int buf, auto hw) {
        TC* Ad = As + buf * GK * LDA_;
        TC* Bd = Bs + buf * GK * LDB_;
        if constexpr (decltype(hw)::value == 1) {
#pragma unroll
            for (int h = 0; h < 2 * NPB; ++h) rb[h] *= rw[h];
        }
        if constexpr (decltype(hw)::value == 2) {
#pragma unroll
            for (int h = 0; h < 2 * NPB; ++h) rb[h] = rw[h] - rb[h];
        }
#pragma unroll
        for (int h = 0; h < NPA; ++h) {
            const int e = tid + 256 * h, kx = e / (HM / 2), xx = 2 * (e % (HM / 2));
            const int kk = 2 * (tid & 7), xk = (tid >> 3) + 32 * h;
            if (TA) {
                *reinterpret_cast<TC2*>(&Ad[kx * LDA_ + xx]) = TC2{ra[2 * h], ra[2 * h + 1]};
            } else {
                Ad[kk * LDA_ + xk] = ra[2 * h]; Ad[(kk + 1) * LDA_ + xk] = ra[2 * h + 1];
            }
        }
#pragma unroll
        for (int h = 0; h < NPB; ++h) {
            const int e = tid + 256 * h, kx = e / (HN / 2), xx = 2 * (e % (HN / 2));
            const int kk = 2 * (tid & 7), xk = (tid >> 3) + 32 * h;
            if (TB) {
                Bd[kk * LDB_ + xk] = rb[2 * h]; Bd[(kk + 1) * LDB_ + xk] = rb[2 * h + 1];
            } else {
                *reinterpret_cast<TC2*>(&Bd[kx * LDB_ + xx]) = TC2{rb[2 * h], rb[2 * h + 1]};
            }
        }
    };
    // Three panels in flight: panel p is multiplied out of LDS buffer `cur`; panel p + 1 (fetched into registers during
    // panel p - 1) is written to the other LDS buffer after the first of the four MFMA groups has been issued, so that
    // the LDS stores complete under the remaining three; panel p + 2 then starts its global loads, a full panel of
    // MFMAs ahead of its use -- the barrier at the end of a panel waits neither for memory nor for the LDS.  The operand
    // fragments of MFMA group s + 1 are read from LDS before group s issues.  (61.8 -> 64.6 TFLOP/s on exact tiles.)
    const auto W1 = std::integral_constant<int, 1>{};
    const auto W0 = std::integral_constant<int, 0>{};
    const auto W2 = std::integral_constant<int, 2>{};
    fetch(klo);
    if (wk) stage(0, W1); else if (Bs2) stage(0, W2); else stage(0, W0);
    if (klo + GK < khi) fetch(klo + GK);
    __syncthreads();
    int cur = 0;
    TC av[2][WM], bv[2][WN];
    {
        const TC* Ab = As + wi + r;
        const TC* Bb = Bs + wj + r;
#pragma unroll
        for (int a = 0; a < WM; ++a) av[0][a] = Ab[q * LDA_ + 16 * a];
#pragma unroll
        for (int b = 0; b < WN; ++b) bv[0][b] = Bb[q * LDB_ + 16 * b];
    }
    // The one barrier of a panel sits BEFORE its last MFMA group, not after it: by then every wave has issued its last
    // fragment reads of buffer `cur` (group s + 1 is read before group s issues) and the LDS stores of panel p + 1 (issued
    // after group 0) are long complete, so the first fragments of panel p + 1 are read from the other buffer under the last
    // MFMA group and the next panel starts with its operands in registers -- no LDS latency and no barrier wait at the panel
    // boundary with an empty matrix pipe.
    int k0 = klo;
    // Steady state (panels k0 + GK and k0 + 2 GK exist and are full): one basic block per panel, so that
    // the LDS stores of panel p + 1 and the global loads of panel p + 2 can be placed BETWEEN the 16 MFMAs of group 0 instead
    // of after them (sched_group_barrier pipeline: an f64 MFMA occupies the matrix pipe for 64 cycles, room for three or four
    // other instructions of the same wave) -- in the generic loop below they sit in their own blocks behind uniform branches
    // and the matrix pipe of this wave idles for their ~100 instructions.
    auto steady = [&](auto hw) {
        while (k0 + 3 * GK <= khi) {
            const TC* Ab = As + cur * GK * LDA_ + wi + r;
            const TC* Bb = Bs + cur * GK * LDB_ + wj + r;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < WM; ++a) av[1][a] = Ab[(4 + q) * LDA_ + 16 * a];
#pragma unroll
            for (int b = 0; b < WN; ++b) bv[1][b] = Bb[(4 + q) * LDB_ + 16 * b];
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) acc[a][b] = MF::mma(av[0][a], bv[0][b], acc[a][b]);
            stage(cur ^ 1, hw);
            fetch_fast(k0 + 2 * GK, hw);
            __builtin_amdgcn_sched_group_barrier(0x100, WM + WN, 0);                 // the fragment reads of group 1
#pragma unroll
            for (int i = 0; i < WM * WN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA
                if (i < WM * WN / 2) {
                    __builtin_amdgcn_sched_group_barrier(0x200, PIPE_DSW, 0);        // LDS stores of panel p + 1
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x002, PIPE_VALU, 0);       // address arithmetic
                    __builtin_amdgcn_sched_group_barrier(0x020, PIPE_VMEM, 0);       // global loads of panel p + 2
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 1; st < GK / 4; ++st) {
                if (st + 1 < GK / 4) {
#pragma unroll
                    for (int a = 0; a < WM; ++a) av[(st + 1) & 1][a] = Ab[(4 * (st + 1) + q) * LDA_ + 16 * a];
#pragma unroll
                    for (int b = 0; b < WN; ++b) bv[(st + 1) & 1][b] = Bb[(4 * (st + 1) + q) * LDB_ + 16 * b];
                } else {
                    __syncthreads();
                    const TC* An = As + (cur ^ 1) * GK * LDA_ + wi + r;
                    const TC* Bn = Bs + (cur ^ 1) * GK * LDB_ + wj + r;
#pragma unroll
                    for (int a = 0; a < WM; ++a) av[0][a] = An[q * LDA_ + 16 * a];
#pragma unroll
                    for (int b = 0; b < WN; ++b) bv[0][b] = Bn[q * LDB_ + 16 * b];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int b = 0; b < WN; ++b) acc[a][b] = MF::mma(av[st & 1][a], bv[st & 1][b], acc[a][b]);
                __builtin_amdgcn_sched_barrier(0);
            }
            cur ^= 1;
            k0 += GK;
        }
    };
    if (GEMM_PIPE && tile_fast) {
        if (wk) steady(W1); else if (Bs2) steady(W2); else steady(W0);
    }
    for (; k0 < khi; k0 += GK) {
        const TC* Ab = As + cur * GK * LDA_ + wi + r;
        const TC* Bb = Bs + cur * GK * LDB_ + wj + r;
        const bool more = k0 + GK < khi;
#pragma unroll
        for (int st = 0; st < GK / 4; ++st) {
            if (st + 1 < GK / 4) {
#pragma unroll
                for (int a = 0; a < WM; ++a) av[(st + 1) & 1][a] = Ab[(4 * (st + 1) + q) * LDA_ + 16 * a];
#pragma unroll
                for (int b = 0; b < WN; ++b) bv[(st + 1) & 1][b] = Bb[(4 * (st + 1) + q) * LDB_ + 16 * b];
            } else {
                __syncthreads();
                if (more) {
                    const TC* An = As + (cur ^ 1) * GK * LDA_ + wi + r;
                    const TC* Bn = Bs + (cur ^ 1) * GK * LDB_ + wj + r;
#pragma unroll
                    for (int a = 0; a < WM; ++a) av[0][a] = An[q * LDA_ + 16 * a];
#pragma unroll
                    for (int b = 0; b < WN; ++b) bv[0][b] = Bn[q * LDB_ + 16 * b];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) acc[a][b] = MF::mma(av[st & 1][a], bv[st & 1][b], acc[a][b]);
            __builtin_amdgcn_sched_barrier(0);
            if (st == 0) {
                if (more) { if (wk) stage(cur ^ 1, W1); else if (Bs2) stage(cur ^ 1, W2); else stage(cur ^ 1, W0); }
                if (k0 + 2 * GK < khi) fetch(k0 + 2 * GK);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        cur ^= 1;
    }
    const bool has_beta = g.beta != real(0);
    // Mirrored store (tri & 16): a diagonal tile stores its lower half and mirrors it too -- with contraction weights on the B side
    // the two halves of a diagonal tile are a_i (w a_j) and a_j (w a_i), equal only up to rounding, and every consumer (and the packed
    // exchange between ranks) relies on the stored matrix being symmetric BIT FOR BIT
    // (tests/test_gpu_large_split.py::test_exchanged_symmetric_blocks_are_exactly_symmetric_in_memory).
    const bool sym_diag = (g.tri & 16) && i0 == j0;
    if (g.epi_on) {
        // extended epilogue: an extra matrix E and a diagonal term on the output, and an optional second output with its own
        // coefficients -- K + c S + jI next to S, A + jI next to A, P^T - K Aji: each used to be a pass over an (L, m, m) array
        const TS* __restrict__ E = g.E ? static_cast<const TS*>(g.E) + (size_t)l * g.se : nullptr;
        TS* C2 = g.C2 ? static_cast<TS*>(g.C2) + (size_t)l * g.sc2 : nullptr;
        const real alpha = g.alpha_dev ? g.alpha * *g.alpha_dev : g.alpha;
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int b = 0; b < WN; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int gi = i0 + wi + a * 16 + MF::row(q, e), gj = j0 + wj + b * 16 + r;
                    if (gi < g.M && gj < g.N && !(sym_diag && gi < gj)) {
                        const real av = (real)acc[a][b][e], dd = gi == gj ? real(1) : real(0);
                        const size_t o = (size_t)gi * g.ldc + gj;
                        const real ev = E ? (real)E[(size_t)gi * g.lde + gj] : real(0);
                        C[o] = (TS)(alpha * av + (has_beta ? g.beta * (real)C[o] : real(0)) + g.g1 * ev + g.d1 * dd);
                        if (C2) C2[(size_t)gi * g.ldc2 + gj] = (TS)(g.a2 * av + g.g2 * ev + g.d2 * dd);
                        if ((g.tri & 16) && gi != gj) {          // mirror: a below-diagonal tile, or the lower half of a diagonal tile
                            const size_t oT = (size_t)gj * g.ldc + gi;
                            const real evT = E ? (g.e_sym ? ev : (real)E[(size_t)gj * g.lde + gi]) : real(0);
                            C[oT] = (TS)(alpha * av + (has_beta ? g.beta * (real)C[oT] : real(0)) + g.g1 * evT);
                            if (C2) C2[(size_t)gj * g.ldc2 + gi] = (TS)(g.a2 * av + g.g2 * evT);
                        }
                    }
                }
        return;
    }
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int gi = i0 + wi + a * 16 + MF::row(q, e), gj = j0 + wj + b * 16 + r;
                if (gi < g.M && gj < g.N && !(sym_diag && gi < gj)) {
                    const size_t o = (size_t)gi * g.ldc + gj;
                    C[o] = (TS)(g.alpha * (real)acc[a][b][e] + (has_beta ? g.beta * (real)C[o] : real(0)));
                    if ((g.tri & 16) && gi != gj) {          // mirror: a below-diagonal tile, or the lower half of a diagonal tile
                        const size_t oT = (size_t)gj * g.ldc + gi;
                        C[oT] = (TS)(g.alpha * (real)acc[a][b][e] + (has_beta ? g.beta * (real)C[oT] : real(0)));
                    }
                }
            }
}

template <bool TA, bool TB, int WT, typename TS, typename TC, bool VEC>
__global__ __launch_bounds__(256, 2) void k_gemm_batched(GemmArgs g) {
    extern __shared__ __align__(16) unsigned char hs_raw[];
    // XCD-aware order: hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own L2);
    // remapping id -> (id % 8) * ceil(total / 8) + id / 8 gives every XCD a contiguous run of tiles, i.e. whole
    // matrices of the batch, so the operand panels a tile row / column shares are fetched into ONE L2
    const int total = g.tiles_n * g.tiles_m * g.batch, per = (total + 7) / 8;
    const int lid = g.xcd_remap ? (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    if (lid >= total) return;
    const int l = lid / (g.tiles_n * g.tiles_m), tt = lid % (g.tiles_n * g.tiles_m);
    gemm_tile<TA, TB, WT, WT, TS, TC, VEC>(g, l, (tt / g.tiles_n) * 32 * WT, (tt % g.tiles_n) * 32 * WT,
                                           reinterpret_cast<TC*>(hs_raw));
}

// Rectangular tiles 32 WM x 32 WN: twice the workgroups of the next larger square tile for problems that would otherwise put one
// workgroup on a CU (256^3 x 16 with 64-tiles, 1024 x 256 x 256 x 16 with 128-tiles: config 3's products)
template <bool TA, bool TB, int WM, int WN, typename TS, typename TC, bool VEC>
__global__ __launch_bounds__(256, 2) void k_gemm_rect(GemmArgs g) {
    extern __shared__ __align__(16) unsigned char hs_raw[];
    const int total = g.tiles_n * g.tiles_m * g.batch, per = (total + 7) / 8;
    const int lid = g.xcd_remap ? (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    if (lid >= total) return;
    const int l = lid / (g.tiles_n * g.tiles_m), tt = lid % (g.tiles_n * g.tiles_m);
    gemm_tile<TA, TB, WM, WN, TS, TC, VEC>(g, l, (tt / g.tiles_n) * 32 * WM, (tt % g.tiles_n) * 32 * WN,
                                           reinterpret_cast<TC*>(hs_raw));
}

// Mixed tiling for extents that are not multiples of 128: full 128 x 128 tiles in the interior and ONE row / column of
// 32 E-wide edge tiles (E = 1, 2: remainders up to 32 / 64; larger remainders take a bounds-checked full tile) in the same
// launch -- m = 800 = 6 x 128 + 32 runs 36 interior tiles at the large-tile rate and pads 13 thin ones instead of padding
// every tile row and column (128-tiles: 25 % padding, 64-tiles: 8 % at the lower small-tile rate).
// E = 3: an extent is cut into a tiles of 128 and b tiles of 96 (every multiple of 32 from 192 on is 128 a + 96 b exactly:
// 800 = 4 x 128 + 3 x 96), so EVERY tile is a large one.  A 128 x 32 edge tile costs about as much as a 128 x 128 tile (4
// MFMAs per wave between the same staging, loads and barrier of a k-panel): 512 x 768 x 800 x 64 ran 66.1 TFLOP/s and
// 512 x 800 x 800 x 64 56.9, i.e. the 32 extra columns cost 21 % more time for 4 % more flops.
template <bool TA, bool TB, int E, typename TS, typename TC, bool VEC>
__global__ __launch_bounds__(256, 2) void k_gemm_mixed(GemmArgs g) {
    extern __shared__ __align__(16) unsigned char hs_raw[];
    TC* hs = reinterpret_cast<TC*>(hs_raw);
    const int total = g.tiles_n * g.tiles_m * g.batch, per = (total + 7) / 8;
    const int lid = g.xcd_remap ? (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    if (lid >= total) return;
    // (the edge tiles stay interleaved in row-major tile order: collected at the end of the launch they run together, all of
    // them at the low arithmetic intensity of a thin tile -- measured 43 instead of 53 TFLOP/s at 800^3 x 64)
    const int l = lid / (g.tiles_n * g.tiles_m), tt = lid % (g.tiles_n * g.tiles_m);
    const int ti = tt / g.tiles_n, tj = tt % g.tiles_n;
    const bool em = ti >= g.full_m, en = tj >= g.full_n;       // (E = 1, 2: at most the last tile row / column; E = 3: several)
    const int i0 = em ? g.full_m * 128 + (ti - g.full_m) * 32 * E : ti * 128;
    const int j0 = en ? g.full_n * 128 + (tj - g.full_n) * 32 * E : tj * 128;
    if (!em && !en) gemm_tile<TA, TB, 4, 4, TS, TC, VEC>(g, l, i0, j0, hs);
    else if (!em) gemm_tile<TA, TB, 4, E, TS, TC, VEC>(g, l, i0, j0, hs);
    else if (!en) gemm_tile<TA, TB, E, 4, TS, TC, VEC>(g, l, i0, j0, hs);
    else gemm_tile<TA, TB, E, E, TS, TC, VEC>(g, l, i0, j0, hs);
}

// (Tried and removed, round 2: a persistent form whose LDS double buffer keeps running across tile boundaries -- the first
// panel of the next tile in flight under the last MFMAs of the current one, C stores draining under the next tile.  800^3 x
// 64: 41.8 vs 42.4 TFLOP/s, 2048^3 x 16: 58.2 vs 58.7 -- the per-workgroup prologue / epilogue is not where the 32 % of
// idle matrix-pipe time goes.  Also without effect: 4 instead of 2 workgroups per CU for the 64-tiles.  PMC at 800^3 x 64:
// matrix pipe busy 68 %, LDS busy 30 % (14 % of it bank conflicts of the transposing stores), L2 hit rate 72 %, waves parked
// at waitcnt / barrier 19 % of their lifetime; the B-operand layout [j][k] (tb = 1) runs 46 vs 42 TFLOP/s for [k][j].)

// ---------------------------------------------------------------------------------------------
// Blocked Gauss-Jordan inverse (no pivoting; SPD inputs), block size 32, for m < SVGP_CHOL_INVERSE_MIN_M: the pivot blocks
// are inverted by a single-wave register sweep (gj32_sweep), the block steps run as ONE launch each (k_bgjf_step below).
// Rows / columns >= m of the last block behave as an identity pad.  From SVGP_CHOL_INVERSE_MIN_M on the inverse comes
// from the Cholesky factor (cholesky.hip).  (Rounds 1-2 also had a three-launch-per-step form and a two-level form with
// 128-wide outer steps for 512 <= m < 640; potrf + potri is faster there since its triangular inverse went to the MFMA:
// 727 against 905 us at 512 x 16.  What that work established stays true for any wider block step: the textbook form --
// explicit P^-1 A[kb][:] and A[:, kb] P^-1 products with a 128 x 128 pivot inverse -- loses the residual |A X - I| by
// cond(P_128) on the K + jitter matrices of this model, 3e-4 .. 3e-2 at jitter 1e-6 where the 32-block sweep gives 8e-9.)
// ---------------------------------------------------------------------------------------------
#define NB 32

__device__ __forceinline__ real wave_sum_la(real x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}

// Wave 0 of the workgroup inverts the 32 x 32 block staged in LDS (single-wave in-register Gauss-Jordan, sweep32.hpp: 4 x 16
// lanes of 8 x 2 register blocks, pivot column by DPP row broadcast), writes P^-1 to `Pinv` (and `Pcap`), adds log det to
// logdet[l] (sets it when `first`).
// `ld_prev`: the log det accumulated so far (0 for the first block), read by the CALLER at the top of its workgroup -- read here, behind
// the sweep, it was a global round trip at the very end of every block step's chain.
__device__ __forceinline__ void gj32_sweep(const real (*P)[NB + 1], real* __restrict__ Pinv, real* __restrict__ Pcap,
                                           real* __restrict__ logdet_l, real ld_prev) {
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    const real mypiv = sweep32::gauss_jordan_32(
        lane, NB, [&](int i, int j) { return P[i][j]; },
        [&](int i, int j, real v) {
            Pinv[i * NB + j] = v;
            if (Pcap) Pcap[i * NB + j] = v;
        });
    const real lg = wave_sum_la(log(mypiv));
    if (lane == 0) *logdet_l = ld_prev + lg;
}

// ---------------------------------------------------------------------------------------------
// The 32-block sweep with ONE launch per block step.  A three-kernel form
// needs a row-panel launch before the trailing update (every trailing tile reads the scaled pivot row block) and a
// copy of the old pivot column; here a tile recomputes the scaled pivot row block it needs (one extra 32^3 product)
// and the step reads buffer X and writes buffer Y (ping-pong between the matrix and the workspace), so nothing is
// read after it was overwritten.  Same arithmetic per element, 1 + m/32 launches instead of 1 + 2 m/32: at m = 256
// the sweep is a chain of dependent ~10 us launches, not flops.  A second matrix set (Xe / Ye) rides in the same
// launches: the (K_mm + jI) inverse next to the L channel matrices (K_mm + c S_l + jI).
// ---------------------------------------------------------------------------------------------
struct BgjfArgs {
    int m, kb, batch, nmain;
    const real* X; real* Y;        // (nmain, m, m)
    const real* Xe; real* Ye;      // (batch - nmain, m, m)
    real* Pinv;                    // (2, batch, 32, 32)
    real* logdet; real* logdet_e;  // (nmain), (batch - nmain)
};
__device__ __forceinline__ real bgjf_get(const real* X, int m, int bi, int bj, int i, int j, bool pivot) {
    const int gi = bi * NB + i, gj = bj * NB + j;
    return (gi < m && gj < m) ? X[(size_t)gi * m + gj] : ((pivot && i == j) ? real(1) : real(0));
}
__global__ __launch_bounds__(256) void k_bgjf_pivot0(BgjfArgs g) {
    __shared__ real P[NB][NB + 1];
    const int l = blockIdx.x;
    const real* X = l < g.nmain ? g.X + (size_t)l * g.m * g.m : g.Xe + (size_t)(l - g.nmain) * g.m * g.m;
    for (int t = threadIdx.x; t < NB * NB; t += blockDim.x) P[t / NB][t % NB] = bgjf_get(X, g.m, 0, 0, t / NB, t % NB, true);
    __syncthreads();
    gj32_sweep(P, g.Pinv + (size_t)l * NB * NB, nullptr, l < g.nmain ? g.logdet + l : g.logdet_e + (l - g.nmain), real(0));
}
// 32 x 32 product on the f64 MFMA: wave w of the 4 owns the 16 x 16 quadrant (w >> 1, w & 1); lane (r = lane & 15,
// q = lane >> 4) receives elements (row 16 (w >> 1) + q + 4 e, column 16 (w & 1) + r), e = 0..3.  A tenth of the LDS
// traffic of the VALU form mm32 (which is LDS-bound: 5 reads per 4 FMAs).
__device__ __forceinline__ d4_t mm32_mfma(const real (*X)[NB + 1], const real (*Y)[NB + 1]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const int wi = 16 * (w >> 1), wj = 16 * (w & 1);
    d4_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int k0 = 0; k0 < NB; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[wi + r][k0 + q], Y[k0 + q][wj + r], acc, 0, 0, 0);
    return acc;
}
// grid (nb * nb * batch): one 32 x 32 tile (bi, bj) of matrix l per workgroup, block step kb.  The workgroups that own
// the NEXT pivot tile (kb + 1, kb + 1) carry the single-wave 32 x 32 sweep (8 us) on top of their tile: they take the
// first `batch` workgroup ids, so that the sweep runs under the other tiles' updates instead of after them (in
// (x, y, z) dispatch order the last matrix's pivot tile started in the last round: 17.7 us per step at m = 256 x 17,
// against 9.3 us for the final step, which has no sweep).
__global__ __launch_bounds__(256) void k_bgjf_step(BgjfArgs g) {
    __shared__ real Pv[NB][NB + 1];
    __shared__ real Rk[NB][NB + 1];
    __shared__ real Ck[NB][NB + 1];
    const int kb = g.kb, m = g.m, nb = (m + NB - 1) / NB, nb2 = nb * nb;
    int bi, bj, l;
    if ((kb + 1) * NB < m) {
        const int id = blockIdx.x;
        if (id < g.batch) {
            l = id; bi = bj = kb + 1;
        } else {
            l = (id - g.batch) / (nb2 - 1);
            int t = (id - g.batch) % (nb2 - 1);
            if (t >= (kb + 1) * nb + kb + 1) ++t;
            bi = t / nb; bj = t % nb;
        }
    } else {
        l = blockIdx.x / nb2; bi = (blockIdx.x % nb2) / nb; bj = blockIdx.x % nb;
    }
    const size_t mo = (size_t)(l < g.nmain ? l : l - g.nmain) * m * m;
    const real* X = (l < g.nmain ? g.X : g.Xe) + mo;
    real* Y = (l < g.nmain ? g.Y : g.Ye) + mo;
    const real* Pinv = g.Pinv + ((size_t)(kb & 1) * g.batch + l) * NB * NB;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const int ti = 16 * (w >> 1) + q, tj = 16 * (w & 1) + r;      // this thread's elements: rows ti + 4 e, column tj
    real ld_prev = 0;                                              // pivot workgroup: the log det so far, fetched now for the sweep's end
    if ((kb + 1) * NB < m && (int)blockIdx.x < g.batch && threadIdx.x == 0) ld_prev = l < g.nmain ? g.logdet[l] : g.logdet_e[l - g.nmain];
    real xo[4];                                                    // the tile's own old values, in flight under the products
    if (bi != kb && bj != kb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) xo[e] = bgjf_get(X, m, bi, bj, ti + 4 * e, tj, false);
    }
    for (int t = threadIdx.x; t < NB * NB; t += blockDim.x) {
        const int rr = t / NB, c = t % NB;
        Pv[rr][c] = Pinv[t];
        Rk[rr][c] = bj == kb ? Pinv[t] : bgjf_get(X, m, kb, bj, rr, c, false);
        Ck[rr][c] = bi == kb ? real(0) : bgjf_get(X, m, bi, kb, rr, c, false);
    }
    __syncthreads();
    d4_t out;
    if (bj == kb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = Pv[ti + 4 * e][tj];
    } else {
        out = mm32_mfma(Pv, Rk);                         // scaled pivot row block P^-1 X[kb][bj]
    }
    if (bi == kb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gi = kb * NB + ti + 4 * e, gj = bj * NB + tj;
            if (gi < m && gj < m) Y[(size_t)gi * m + gj] = out[e];
        }
        return;
    }
    if (bj != kb) {                                      // (for bj == kb Rk already holds P^-1)
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) Rk[ti + 4 * e][tj] = out[e];
        __syncthreads();
    }
    out = mm32_mfma(Ck, Rk);                             // old column block times the scaled pivot row block
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int gi = bi * NB + ti + 4 * e, gj = bj * NB + tj;
        if (gi < m && gj < m) {
            out[e] = (bj == kb) ? -out[e] : xo[e] - out[e];
            Y[(size_t)gi * m + gj] = out[e];
        } else {
            out[e] = (gi - bi * NB == gj - bj * NB) ? real(1) : real(0);     // identity pad (read below when pivot)
        }
    }
    // look-ahead: this workgroup just produced the NEXT pivot block -> invert it now
    if (bi == kb + 1 && bj == kb + 1 && (kb + 1) * NB < m) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) Pv[ti + 4 * e][tj] = out[e];
        __syncthreads();
        gj32_sweep(Pv, g.Pinv + ((size_t)((kb + 1) & 1) * g.batch + l) * NB * NB, nullptr,
                   l < g.nmain ? g.logdet + l : g.logdet_e + (l - g.nmain), ld_prev);
    }
}

}  // namespace

// prec 0: float64 storage + arithmetic; 1: float64 storage, float32 MFMA arithmetic; 2: float32 storage + arithmetic
static int gemm_launch(int prec, int tri, int ta, int tb, int M, int N, int K, double alpha, const void* A, int lda,
                       long long strideA, const void* B, int ldb, long long strideB, double beta, void* C, int ldc,
                       long long strideC, int batch, void* stream, const void* wk = nullptr, int ldw = 0,
                       long long strideW = 0, const svgp_gemm_epi* epi = nullptr, const void* bsub = nullptr) {
    SVGP_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 0, SVGP_ERR_INVALID, "negative dimension");
    if (M == 0 || N == 0 || batch == 0) return SVGP_OK;
    SVGP_REQUIRE(A && B && C, SVGP_ERR_INVALID, "NULL device pointer");
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.ta = ta; g.tb = tb; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sa = strideA; g.sb = strideB; g.sc = strideC; g.alpha = alpha; g.beta = beta; g.A = A; g.B = B; g.C = C;
    g.tri = tri;
    g.wk = wk; g.ldw = ldw; g.sw = strideW;
    g.bsub = bsub;
    SVGP_REQUIRE(!(wk && bsub), SVGP_ERR_INVALID, "contraction weights and a B transform together are not supported");
    g.epi_on = 0; g.E = nullptr; g.lde = 0; g.se = 0; g.C2 = nullptr; g.sc2 = 0; g.ldc2 = ldc; g.g1 = g.d1 = g.a2 = g.g2 = g.d2 = 0; g.alpha_dev = nullptr; g.e_sym = 0;
    if (epi) {
        SVGP_REQUIRE(prec != 2, SVGP_ERR_INVALID, "extended GEMM epilogue: float64 storage only");
        SVGP_REQUIRE(epi->E || (epi->g1 == 0 && epi->g2 == 0), SVGP_ERR_INVALID, "extended GEMM epilogue: E is NULL");
        g.epi_on = 1; g.E = epi->E; g.lde = epi->lde; g.se = epi->se; g.C2 = epi->C2; g.sc2 = epi->sc2;
        g.ldc2 = epi->ldc2 > 0 ? epi->ldc2 : ldc;
        g.g1 = epi->g1; g.d1 = epi->d1; g.a2 = epi->a2; g.g2 = epi->g2; g.d2 = epi->d2; g.alpha_dev = epi->alpha_dev; g.e_sym = epi->e_sym;
    }
    //  the mirrored store treats every tile with i0 != j0 as lying strictly below the diagonal: true only for a
    // square output cut identically along rows and columns
    SVGP_REQUIRE(!(tri & 16) || (M == N && (tri & 1)), SVGP_ERR_INVALID, "mirrored store needs a square lower-triangle product");
    const long long blocks128 = (long long)((N + 127) / 128) * ((M + 127) / 128) * batch;
    // tile choice: 128 x 128 tiles run ~9 % faster per useful flop than 64 x 64 (57.6 vs 52.9 TFLOP/s at 2048^3, no padding)
    // but pad M, N up to multiples of 128: at 800 x 800 that is 25 % wasted tile area against 8 % (measured 39.7 vs 41.3-45.6
    // TFLOP/s), so compare padded area / rate; below ~192 large tiles the small ones also fill the chip better
    const double cost4 = (double)((M + 127) / 128) * ((N + 127) / 128) * 16384.0 / 57.6;
    const double cost2 = (double)((M + 63) / 64) * ((N + 63) / 64) * 4096.0 / 52.9;
    // fewer than one 64 x 64 tile per CU: 32 x 32 tiles (4x the workgroups; 256^3 batch 1: 18.3 -> 9.0 us)
    // (lower-triangle products count the tiles that run: config-3 statistics, 256 x 256 x 1024 x 16, are 160 workgroups of 64)
    const long long nt64 = (M + 63) / 64;
    const long long blocks64 = ((tri & 1) && M == N ? nt64 * (nt64 + 1) / 2 : (long long)((N + 63) / 64) * nt64) * batch;
    // (tried: 160 x 160 tiles, 5 x 5 MFMA tiles per wave, no padding at m = 800 -- 512 registers per lane plus spills and one
    // wave per SIMD: 25-38 TFLOP/s at 800^3 x 64 against 41-46 for the 64-tiles; removed)
    int wt = (blocks128 >= 192 && cost4 <= cost2) ? 4 : (blocks64 < 256 ? 1 : 2);
    // mixed tiling (k_gemm_mixed, float64): an extent with a remainder of at most 64 modulo 128 gets 128-tiles + one row / column
    // of 32- or 64-wide edge tiles
    static const int mixed_on = [] { const char* e = getenv("SVGP_GEMM_MIXED"); return (e && e[0] == '0') ? 0 : 1; }();
    const int rem_m = M % 128, rem_n = N % 128;
    const bool edge_m = rem_m > 0 && rem_m <= 64, edge_n = rem_n > 0 && rem_n <= 64;
    // (short contractions -- the K = 128 panel solves / column updates of the blocked Cholesky -- are prologue / epilogue bound and
    // rarely fill two workgroups per CU with 128-tiles: 672 x 128 x 128 x 65 48.2 -> 38.8 us with the 64-tiles, 544: 46.3 -> 34.5;
    // potrf 800 x 65 1.32 -> 1.13 ms, SPRITES m = 800 step 25.45 -> 25.10 ms)
    static const int shortk_on = [] { const char* e = getenv("SVGP_GEMM_SHORTK"); return (e && e[0] == '0') ? 0 : 1; }();
    const bool short_k = shortk_on && K <= 256 && blocks128 < 1024;
    const bool mixed = mixed_on && prec == 0 && M >= 128 && N >= 128 && blocks128 >= 192 && (edge_m || edge_n) && !short_k;
    if (short_k && prec == 0 && wt == 4 && blocks64 >= 512) wt = 2;       // (640 x 128 x 128 x 65: 38.9 -> 37.0 us, 512: 39.0 -> 32.5)
    int edge_w = ((edge_m && rem_m > 32) || (edge_n && rem_n > 32)) ? 2 : 1;
    if (mixed) wt = 4;
    // 128 a + 96 b decomposition of both extents (E = 3) when neither is small: least padding, then most 128-tiles
    static const int t96_on = [] { const char* e = getenv("SVGP_GEMM_T96"); return (e && e[0] == '0') ? 0 : 1; }();
    auto cut = [](int X, int& a, int& b) {
        int best = 1 << 30;
        for (int bb = 0; bb <= 3; ++bb) {
            int aa = X - 96 * bb <= 0 ? 0 : (X - 96 * bb + 127) / 128;
            const int pad = 128 * aa + 96 * bb;
            if (pad >= X && pad < best) { best = pad; a = aa; b = bb; }
        }
    };
    int am = 0, bm = 0, an = 0, bn = 0;
    cut(M, am, bm); cut(N, an, bn);
    const bool t96 = t96_on && mixed_on && prec == 0 && M >= 192 && N >= 192 && blocks128 >= 192 && (bm > 0 || bn > 0);
    if (t96) { wt = 4; edge_w = 3; }
    // rectangular tiles (float64): 128 x 64 where 128-tiles leave fewer than two workgroups per CU, 64 x 32 where 64-tiles do
    static const int rect_on = [] { const char* e = getenv("SVGP_GEMM_RECT"); return (e && e[0] == '0') ? 0 : 1; }();
    int rwm = 0, rwn = 0;
    if (rect_on && prec == 0 && !mixed && !t96 && !(tri & 1)) {
        if (wt == 4 && blocks128 < 512 && M % 128 == 0 && N % 64 == 0) { rwm = 4; rwn = 2; }
        else if (wt == 2 && blocks64 < 512 && M % 64 == 0 && N % 32 == 0) { rwm = 2; rwn = 1; }
    }
    const int ht = 32 * wt;
    const size_t lds = prec == 0 ? (size_t)4 * GK_OF(wt) * (ht + 2) * sizeof(double) : (size_t)4 * GK_OF(wt) * (ht + 16) * sizeof(float);
    g.tiles_n = (N + ht - 1) / ht; g.tiles_m = (M + ht - 1) / ht; g.batch = batch;
    g.full_m = (mixed && edge_m) ? M / 128 : g.tiles_m; g.full_n = (mixed && edge_n) ? N / 128 : g.tiles_n;
    if (rwm) { g.tiles_m = (M + 32 * rwm - 1) / (32 * rwm); g.tiles_n = (N + 32 * rwn - 1) / (32 * rwn); }
    if (t96) { g.full_m = am; g.tiles_m = am + bm; g.full_n = an; g.tiles_n = an + bn; }
    g.xcd_remap = 1;     // measured neutral (+-1 %) at 800^3 x 64 and 2048^3 x 16: the Infinity Cache already absorbs the
                         // cross-XCD panel re-fetches; kept because it never hurts and is the layout the hardware deals
    const long long total = (long long)g.tiles_n * g.tiles_m * batch;
    SVGP_REQUIRE(total < (1LL << 30), SVGP_ERR_UNSUPPORTED, "GEMM grid too large");
    const dim3 grid((unsigned)(g.xcd_remap ? (total + 7) / 8 * 8 : total));
    // 16-byte pair loads: both operands 16-byte aligned (8 for float32 storage) with even leading dimensions / batch strides
    const uintptr_t al = prec == 2 ? 8 : 16;
    const bool vec = ((uintptr_t)A % al) == 0 && ((uintptr_t)B % al) == 0 && lda % 2 == 0 && ldb % 2 == 0 && strideA % 2 == 0 &&
                     strideB % 2 == 0 && ((uintptr_t)bsub % al) == 0;
#define LAUNCH_V(TA_, TB_, WT_, TS_, TC_, V_)                                                                            \
    do {                                                                                                                 \
        SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_batched<TA_, TB_, WT_, TS_, TC_, V_>),    \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
        hipLaunchKernelGGL((k_gemm_batched<TA_, TB_, WT_, TS_, TC_, V_>), grid, dim3(256), lds, (hipStream_t)stream, g);  \
    } while (0)
#define LAUNCH_G(TA_, TB_, WT_, TS_, TC_)                     \
    do {                                                      \
        if (vec) LAUNCH_V(TA_, TB_, WT_, TS_, TC_, true);     \
        else LAUNCH_V(TA_, TB_, WT_, TS_, TC_, false);        \
    } while (0)
#define LAUNCH_T(WT_, TS_, TC_)                                \
    do {                                                       \
        if (ta && tb) LAUNCH_G(true, true, WT_, TS_, TC_);     \
        else if (ta) LAUNCH_G(true, false, WT_, TS_, TC_);     \
        else if (tb) LAUNCH_G(false, true, WT_, TS_, TC_);     \
        else LAUNCH_G(false, false, WT_, TS_, TC_);            \
    } while (0)
#define LAUNCH_P(TS_, TC_)                    \
    do {                                      \
        if (wt == 4) LAUNCH_T(4, TS_, TC_);   \
        else if (wt == 1) LAUNCH_T(1, TS_, TC_); \
        else LAUNCH_T(2, TS_, TC_);           \
    } while (0)
#define LAUNCH_MX(TA_, TB_, E_)                                                                                              \
    do {                                                                                                                 \
        if (vec) {                                                                                                       \
            SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_mixed<TA_, TB_, E_, double, double, true>),  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
            hipLaunchKernelGGL((k_gemm_mixed<TA_, TB_, E_, double, double, true>), grid, dim3(256), lds, (hipStream_t)stream, g); \
        } else {                                                                                                         \
            SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_mixed<TA_, TB_, E_, double, double, false>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
            hipLaunchKernelGGL((k_gemm_mixed<TA_, TB_, E_, double, double, false>), grid, dim3(256), lds, (hipStream_t)stream, g); \
        }                                                                                                                \
    } while (0)
#define LAUNCH_ME(E_)                                  \
    do {                                               \
        if (ta && tb) LAUNCH_MX(true, true, E_);       \
        else if (ta) LAUNCH_MX(true, false, E_);       \
        else if (tb) LAUNCH_MX(false, true, E_);       \
        else LAUNCH_MX(false, false, E_);              \
    } while (0)
#define LAUNCH_RX(TA_, TB_, WM_, WN_)                                                                                        \
    do {                                                                                                                 \
        if (vec) {                                                                                                       \
            SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_rect<TA_, TB_, WM_, WN_, double, double, true>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
            hipLaunchKernelGGL((k_gemm_rect<TA_, TB_, WM_, WN_, double, double, true>), grid, dim3(256), lds, (hipStream_t)stream, g); \
        } else {                                                                                                         \
            SVGP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_rect<TA_, TB_, WM_, WN_, double, double, false>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
            hipLaunchKernelGGL((k_gemm_rect<TA_, TB_, WM_, WN_, double, double, false>), grid, dim3(256), lds, (hipStream_t)stream, g); \
        }                                                                                                                \
    } while (0)
#define LAUNCH_RE(WM_, WN_)                                 \
    do {                                                    \
        if (ta && tb) LAUNCH_RX(true, true, WM_, WN_);      \
        else if (ta) LAUNCH_RX(true, false, WM_, WN_);      \
        else if (tb) LAUNCH_RX(false, true, WM_, WN_);      \
        else LAUNCH_RX(false, false, WM_, WN_);             \
    } while (0)
    if (rwm == 4) LAUNCH_RE(4, 2);
    else if (rwm == 2) LAUNCH_RE(2, 1);
    else if (t96) LAUNCH_ME(3);
    else if (mixed) { if (edge_w == 1) LAUNCH_ME(1); else LAUNCH_ME(2); }
    else if (prec == 0) LAUNCH_P(double, double);
    else if (prec == 1) LAUNCH_P(double, float);
    else LAUNCH_P(float, float);
#undef LAUNCH_P
#undef LAUNCH_ME
#undef LAUNCH_RE
#undef LAUNCH_RX
#undef LAUNCH_MX
#undef LAUNCH_T
#undef LAUNCH_G
#undef LAUNCH_V
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_dgemm_batched(int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda,
                                  long long strideA, const double* B, int ldb, long long strideB, double beta,
                                  double* C, int ldc, long long strideC, int batch, void* stream) {
    return gemm_launch(0, 0, ta, tb, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch, stream);
}
// float64 matrices, products and sums on the float32 MFMA (operands rounded to float32 while staged, float32
// accumulation): the arithmetic of the reference's float32 SPRITES graph on float64 storage
extern "C" int svgp_dgemm_f32c_batched(int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda,
                                       long long strideA, const double* B, int ldb, long long strideB, double beta,
                                       double* C, int ldc, long long strideC, int batch, void* stream) {
    return gemm_launch(1, 0, ta, tb, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch, stream);
}
extern "C" int svgp_sgemm_batched(int ta, int tb, int M, int N, int K, float alpha, const float* A, int lda,
                                  long long strideA, const float* B, int ldb, long long strideB, float beta, float* C,
                                  int ldc, long long strideC, int batch, void* stream) {
    return gemm_launch(2, 0, ta, tb, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch, stream);
}

// C = alpha op(A) op(B) + beta C for a product known to be SYMMETRIC (M == N): only the tiles that touch the lower triangle
// are computed, each below-diagonal tile is stored twice (itself and transposed) -- 45 % fewer tiles at 13 x 13.
// f32c != 0: float32 MFMA arithmetic on the float64 matrices.
int svgp_dgemm_symout_batched(int f32c, int ta, int tb, int M, int K, double alpha, const double* A, int lda, long long strideA,
                              const double* B, int ldb, long long strideB, double beta, double* C, int ldc, long long strideC,
                              int batch, void* stream, const double* wk, int ldw, long long strideW, const svgp_gemm_epi* epi) {
    return gemm_launch(f32c ? 1 : 0, 1 | 16, ta, tb, M, M, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch,
                       stream, wk, ldw, strideW, epi);
}
// C = alpha op(A) (Bsub - op(B)) + beta C: the B operand is transformed while it is staged (Bsub shared by the batch, stored like B)
int svgp_dgemm_bsub_batched(int f32c, int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda, long long strideA,
                            const double* B, int ldb, long long strideB, const double* Bsub, double beta, double* C, int ldc,
                            long long strideC, int batch, void* stream) {
    SVGP_REQUIRE(Bsub, SVGP_ERR_INVALID, "Bsub is NULL");
    return gemm_launch(f32c ? 1 : 0, 0, ta, tb, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch,
                       stream, nullptr, 0, 0, nullptr, Bsub);
}
// the general product with the extended epilogue (svgp_gemm_epi)
int svgp_dgemm_epi_batched(int f32c, int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda, long long strideA,
                           const double* B, int ldb, long long strideB, double beta, double* C, int ldc, long long strideC,
                           int batch, void* stream, const svgp_gemm_epi* epi) {
    return gemm_launch(f32c ? 1 : 0, 0, ta, tb, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch,
                       stream, nullptr, 0, 0, epi);
}

// the float64 GEMM with triangular structure hints (see GemmArgs::tri); tiles / k-panels that the hints exclude are
// skipped, everything else is computed as usual (excluded operand parts must hold zeros where a tile straddles them)
int svgp_dgemm_tri_batched(int tri, int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda,
                           long long strideA, const double* B, int ldb, long long strideB, double beta, double* C, int ldc,
                           long long strideC, int batch, void* stream, const svgp_gemm_epi* epi) {
    return gemm_launch(0, tri, ta, tb, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch, stream,
                       nullptr, 0, 0, epi);
}

// ---- split-K form for one GEMM with few output tiles and a long contraction (the dense layers of the moving-ball
// MLPs: 1050 x 1024 x 500 and their weight gradients with K = 1050 rows).  The contraction is cut into S slices that
// run as the batch dimension of k_dgemm_batched into S partial products in `scratch`; one kernel adds them in fixed
// order: C = alpha * sum_s P_s + beta * C.  Deterministic (no atomics).
namespace {
// rows >= row2 are scaled by alpha2 instead of alpha (two stacked products with their own factors in one contraction)
__global__ void k_splitk_reduce(int M, int N, int S, real alpha, real beta, const real* __restrict__ part,
                                real* __restrict__ C, int ldc, real alpha2, int row2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long MN = (long long)M * N;
    if (i >= MN) return;
    real s = 0;
    for (int k = 0; k < S; ++k) s += part[(size_t)k * MN + i];
    const int row = (int)(i / N);
    real* c = C + (size_t)row * ldc + (i % N);
    *c = (row >= row2 ? alpha2 : alpha) * s + (beta != real(0) ? beta * *c : real(0));
}
// number of K slices: enough workgroups to cover the chip a few times, slices of at least 64
int splitk_slices(int M, int N, int K) {
    const long long tiles = (long long)((M + 63) / 64) * ((N + 63) / 64);
    if (tiles >= 256 || K < 256) return 1;
    long long s = (512 + tiles - 1) / tiles;
    if (s > K / 64) s = K / 64;
    if (s > 32) s = 32;
    return s < 2 ? 1 : (int)s;
}
}  // namespace

extern "C" long long svgp_dgemm_splitk_scratch_elems(int M, int N, int K) {
    if (M < 0 || N < 0 || K < 0) return -1;
    const int S = splitk_slices(M, N, K);
    return S == 1 ? 0 : (long long)(S + 1) * M * N;
}

extern "C" int svgp_dgemm_splitk(int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda,
                                 const double* B, int ldb, double beta, double* C, int ldc, double* scratch,
                                 long long scratch_elems, void* stream) {
    return svgp_dgemm_splitk_rows2(ta, tb, M, N, K, alpha, alpha, M, A, lda, B, ldb, beta, C, ldc, scratch, scratch_elems, stream);
}
// internal: rows [row2, M) of the result take the factor alpha2 (gp_large.hip: ud = a^T Kn and td = c b^T Kn as ONE contraction over
// the rows with the operands stacked)
int svgp_dgemm_splitk_rows2(int ta, int tb, int M, int N, int K, double alpha, double alpha2, int row2, const double* A, int lda,
                            const double* B, int ldb, double beta, double* C, int ldc, double* scratch, long long scratch_elems,
                            void* stream) {
    SVGP_REQUIRE(M >= 0 && N >= 0 && K >= 0, SVGP_ERR_INVALID, "negative dimension");
    if (M == 0 || N == 0) return SVGP_OK;
    const int S = splitk_slices(M, N, K);
    if (S == 1 && row2 >= M) return svgp_dgemm_batched(ta, tb, M, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1, stream);
    if (S == 1) {          // two factors without a reduction pass: the two row ranges as two products
        SVGP_REQUIRE(ta == 1, SVGP_ERR_UNSUPPORTED, "row-split factors without split-K: transposed A only");
        int rc1 = svgp_dgemm_batched(ta, tb, row2, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1, stream);
        if (rc1) return rc1;
        return svgp_dgemm_batched(ta, tb, M - row2, N, K, alpha2, A + row2, lda, 0, B, ldb, 0, beta, C + (size_t)row2 * ldc, ldc, 0, 1,
                                  stream);
    }
    SVGP_REQUIRE(scratch && scratch_elems >= (long long)(S + 1) * M * N, SVGP_ERR_INVALID,
                 "split-K scratch too small: need %lld doubles", (long long)(S + 1) * M * N);
    const int Kc = (K / S) & ~3, rem = K - S * Kc;          // equal slices (multiple of 4) + one remainder slice
    const long long sA = ta ? (long long)Kc * lda : Kc, sB = tb ? Kc : (long long)Kc * ldb, MN = (long long)M * N;
    int rc = svgp_dgemm_batched(ta, tb, M, N, Kc, 1.0, A, lda, sA, B, ldb, sB, 0.0, scratch, N, MN, S, stream);
    if (rc) return rc;
    int nparts = S;
    if (rem > 0) {
        rc = svgp_dgemm_batched(ta, tb, M, N, rem, 1.0, A + (size_t)S * sA, lda, 0, B + (size_t)S * sB, ldb, 0, 0.0,
                                scratch + (size_t)S * MN, N, 0, 1, stream);
        if (rc) return rc;
        nparts = S + 1;
    }
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, N, nparts,
                       alpha, beta, scratch, C, ldc, alpha2, row2);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" size_t svgp_potrf_workspace_elems(int m, int batch);
extern "C" size_t svgp_potri_workspace_elems(int m, int batch);
extern "C" int svgp_potrf_batched(int m, int batch, double* A, int lda, long long strideA, double* logdet, double* work,
                                  void* stream);
extern "C" int svgp_potri_batched(int m, int batch, double* A, const double* linv_blocks, double* work, void* stream);
// From SVGP_CHOL_INVERSE_MIN_M on the inverse is formed from the Cholesky factor (cholesky.hip): half the flops of the
// elimination (m^3 against 2 m^3), nearly all of them in MFMA GEMMs.  Measured (tools/inverse_probe.py, float64, us):
// 256 x 17: fused Gauss-Jordan 150 / potrf + potri 305;  512 x 16: (two-level Gauss-Jordan 905) / 727;  800 x 65: 2714.
#define CHOL_INVERSE_MIN_M SVGP_CHOL_INVERSE_MIN_M

extern "C" size_t svgp_spd_inverse_workspace_elems(int m, int batch) {
    if (m >= CHOL_INVERSE_MIN_M) return svgp_potrf_workspace_elems(m, batch) + svgp_potri_workspace_elems(m, batch);
    return (size_t)batch * (2 * NB * NB + (size_t)m * m);     // pivots + the ping-pong copy
}

// The inverse of a symmetric matrix, made exactly symmetric: Y = (X + X^T) / 2 per 32 x 32 tile pair (X == Y allowed: a
// workgroup owns both tiles of its pair).  The elimination leaves an antisymmetric rounding residue E (1e-9 relative
// at cond 1e7); first-order terms like K E K or Ki A E + E A Ki are antisymmetric too and cancel in every quadratic
// form downstream -- unless a product is computed on its lower triangle and mirrored, which folds them into the
// symmetric part (config-3 shape, jitter 1e-2: `d` moved by 4e-5 against 5e-10 for a one-ulp input perturbation).
__global__ __launch_bounds__(256) void k_symmetrize(int m, int nmain, const real* X, real* Y, const real* Xe, real* Ye) {
    const int ti = blockIdx.y, tj = blockIdx.x, bt = blockIdx.z;
    if (ti > tj) return;
    const size_t mm = (size_t)m * m;
    const real* src = bt < nmain ? X + (size_t)bt * mm : Xe + (size_t)(bt - nmain) * mm;
    real* dst = bt < nmain ? Y + (size_t)bt * mm : Ye + (size_t)(bt - nmain) * mm;
    __shared__ real U[NB][NB + 1], V[NB][NB + 1];
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int r = r0 + 8 * h;
        const int ui = ti * NB + r, uj = tj * NB + c, vi = tj * NB + r, vj = ti * NB + c;
        U[r][c] = (ui < m && uj < m) ? src[(size_t)ui * m + uj] : real(0);
        V[r][c] = (vi < m && vj < m) ? src[(size_t)vi * m + vj] : real(0);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int r = r0 + 8 * h;
        const int ui = ti * NB + r, uj = tj * NB + c, vi = tj * NB + r, vj = ti * NB + c;
        if (ui < m && uj < m) dst[(size_t)ui * m + uj] = real(0.5) * (U[r][c] + V[c][r]);
        if (ti != tj && vi < m && vj < m) dst[(size_t)vi * m + vj] = real(0.5) * (V[r][c] + U[c][r]);
    }
}

// Fused one-launch-per-block-step sweep (m < SVGP_CHOL_INVERSE_MIN_M) over `nmain` matrices at A plus `nextra` at Ae; work
// holds svgp_spd_inverse_workspace_elems(m, nmain + nextra) doubles.
int svgp_spd_inverse_fused(int m, int nmain, double* A, double* logdet, int nextra, double* Ae, double* logdet_e,
                           double* work, void* stream) {
    const int batch = nmain + nextra, nb = (m + NB - 1) / NB;
    const size_t mm = (size_t)m * m;
    BgjfArgs g;
    g.m = m; g.batch = batch; g.nmain = nmain; g.Pinv = work; g.logdet = logdet; g.logdet_e = logdet_e;
    real* W = work + (size_t)batch * 2 * NB * NB;
    real* We = W + (size_t)nmain * mm;
    g.X = A; g.Xe = Ae; g.kb = 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bgjf_pivot0, dim3(batch), dim3(256), 0, s, g);
    SVGP_LAUNCH_CHECK();
    for (int kb = 0; kb < nb; ++kb) {
        g.kb = kb;
        const bool fwd = (kb & 1) == 0;             // even steps read the matrices and write the workspace copy
        g.X = fwd ? A : W; g.Y = fwd ? W : A;
        g.Xe = fwd ? Ae : We; g.Ye = fwd ? We : Ae;
        hipLaunchKernelGGL(k_bgjf_step, dim3((unsigned)nb * nb * batch), dim3(256), 0, s, g);
        SVGP_LAUNCH_CHECK();
    }
    // the result sits in the workspace copy after an odd number of steps; either way it lands in A / Ae symmetrised
    const bool inW = (nb & 1) != 0;
    hipLaunchKernelGGL(k_symmetrize, dim3(nb, nb, batch), dim3(256), 0, s, m, nmain, inW ? W : A, A, inW ? We : Ae, Ae);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// A (batch, m, m) SPD, contiguous -> inverse in place; logdet (batch).  work: svgp_spd_inverse_workspace_elems.
extern "C" int svgp_spd_inverse_batched(int m, int batch, double* A, double* logdet, double* work, void* stream) {
    SVGP_REQUIRE(m >= 1 && batch >= 0, SVGP_ERR_INVALID, "bad m / batch");
    if (batch == 0) return SVGP_OK;
    SVGP_REQUIRE(A && logdet && work, SVGP_ERR_INVALID, "NULL device pointer");
    if (m < CHOL_INVERSE_MIN_M) return svgp_spd_inverse_fused(m, batch, A, logdet, 0, nullptr, nullptr, work, stream);
    int rc = svgp_potrf_batched_band(m, batch, A, m, (long long)m * m, logdet, work, stream);
    if (rc) return rc;
    return svgp_potri_batched_wide(m, batch, A, work, work + svgp_potrf_workspace_elems(m, batch), stream);
}
