// Large-m (m > SVGP_M_MAX) implementation of the GP stages: the m x m matrices live in the workspace
// (HBM / Infinity-Cache resident) and every contraction is a batched float64 MFMA GEMM
// (linalg.hip); inverses are blocked Gauss-Jordan.  Same mathematics, same workspace fields and the
// same exchange blocks (statA / statB) as the LDS-resident path in gp_kernels.hip, so the phases,
// data parallelism and parity tests are unchanged.  Reference lines: SVGPVAE_model.py:220-343.
//
// Round 4, "W form" (restated in oracle/staged_gp.py, gp_*_w): the L3 integrand's k_n^T Ki A_l Ki k_n (SVGPVAE_model.py:281-284)
// is evaluated as w_n^T Si_l w_n with the channel-INDEPENDENT rows w_n = K Ki k_n, W = (Kn Ki) K (A_l = K Si_l K; Ki K is
// not simplified to I).  Same algebra, but the per-channel m^3 products A Ki, Ki A Ki (forward), S Ki, Ki S Ki,
// S Ki A (reverse) and the row product Kn M2_l are gone: 5 of the 13 full 800^3 x 64 products of the SPRITES step.  In
// their place: [Kn; W] Si_l as ONE row product, the statistic SW_l = W^T diag(p_l) W = P^T S_l P beside the reverse statistic
// (Sibar_l gets A2_l - g3/2 SW_l; formed early, from forward quantities: over the rows on one GPU when that is cheaper, from the
// all-reduced S_l when the batch is sharded over ranks), and channel-independent m x m matrices: P^T = K Ki, Pbar = Kn^T Wbar,
// Qs = Kn^T diag(qbar) Kn.  Row sums that enter the gradient of K_mm linearly
// (Pbar, Qs) stay rank-local under data parallelism -- the ranks' shares of Kbar add up in the gradient all-reduce -- so
// cfg.rep_weight is applied HERE to the replicated part of Kbar and the kernel-matrix reverse pass takes Kbar as it is.
#include "common.hpp"

extern "C" int svgp_dgemm_batched(int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda,
                                  long long strideA, const double* B, int ldb, long long strideB, double beta,
                                  double* C, int ldc, long long strideC, int batch, void* stream);
extern "C" int svgp_dgemm_f32c_batched(int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda,
                                       long long strideA, const double* B, int ldb, long long strideB, double beta,
                                       double* C, int ldc, long long strideC, int batch, void* stream);
extern "C" int svgp_spd_inverse_batched(int m, int batch, double* A, double* logdet, double* work, void* stream);

namespace {

__device__ __forceinline__ real gradKL(int flags, int L, const real* state) { return svgp_seed_T(flags, L, state); }

// ---- element-wise / reduction kernels ---------------------------------------------------------
// weights of the statistics.  mode 0: w = 1/s2, a = y/s2.  mode 1: g_pv, g_pm, mvbar (stored), b = c g_pm.
__global__ void k_big_weights(int n_el, int L, int mode, int geco, int clip_pv, real c, const real* __restrict__ state,
                              const real* __restrict__ y, const real* __restrict__ s2,
                              const real* __restrict__ p_m, const real* __restrict__ p_v,
                              const real* __restrict__ e, const real* __restrict__ eps,
                              const real* __restrict__ zbar, real* __restrict__ w, real* __restrict__ a,
                              real* __restrict__ bv, real* __restrict__ stk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_el) return;
    const real p = recip_no_nan(s2[i]);
    if (mode == 0) {
        w[i] = p; a[i] = p * y[i];
    } else {
        const real gT = gradKL(geco, L, state), zb = zbar[i];
        const real pv = p_v[i];
        const real gpv = svgp_gpv(clip_pv, gT, p, zb, eps[i], pv);
        const real gpm = gT * p * (p_m[i] - y[i]) + zb;
        const real g3 = svgp_seed_3(geco, gT);
        const real av = g3 * p * e[i];
        w[i] = gpv; bv[i] = gpm; a[i] = av;                  // g_pv, g_pm, mvbar buffers
        if (stk) {                                           // and [mvbar | g_pm] side by side, (b, 2 L): one stacked operand
            const int n = i / L, l = i % L;
            stk[(size_t)n * 2 * L + l] = av;
            stk[(size_t)n * 2 * L + L + l] = gpm;
        }
    }
}
// (+ the device scalar -g3/2 for the epilogue of the SW product that follows: X0 = A2 - g3/2 SW, see svgp_big_stats)
__global__ void k_big_recip(int n_el, const real* __restrict__ s2, real* __restrict__ p, int flags, int L,
                            const real* __restrict__ state, real* __restrict__ mhalf_g3) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_el) p[i] = recip_no_nan(s2[i]);
    if (i == 0 && mhalf_g3) *mhalf_g3 = real(-0.5) * svgp_seed_3(flags, gradKL(flags, L, state));
}
// The rank-local row terms of the reverse pass in one pass over the rows, X (b, 2 m) = [qbar * Kn | Wbar]:
//   qbar_n = sum_l (g3/2 p_nl - g_pv_nl): the weight of k_n k_n^T in the gradient of Ki (q_n = k^T Ki k inside d and p_v);
//   Wbar[n][j] = -g3 sum_l p_nl (W Si_l)[n][j]: the gradient of the rows W = Kn Ki K (channel sum: W is channel-independent).
// [Qs; Pbar^T] = X^T Kn is then ONE contraction over the rows (svgp_big_stats).
__global__ void k_big_rowlocal(int b, int m, int L, int geco, const real* __restrict__ state, const real* __restrict__ s2,
                               const real* __restrict__ g_pv, const real* __restrict__ Kn, const real* __restrict__ WSi,
                               long long sW, real* __restrict__ X, real* __restrict__ qbar) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)b * m) return;
    const int n = (int)(i / m), j = (int)(i % m);
    const real g3 = svgp_seed_3(geco, gradKL(geco, L, state));
    real s = 0, q = 0;
    for (int l = 0; l < L; ++l) {
        const real p = recip_no_nan(s2[(size_t)n * L + l]);
        s += p * WSi[(size_t)l * sW + i];
        q += real(0.5) * g3 * p - g_pv[(size_t)n * L + l];
    }
    X[(size_t)n * 2 * m + j] = q * Kn[i];
    X[(size_t)n * 2 * m + m + j] = -g3 * s;
    if (j == 0) qbar[n] = q;
}
// out[l] = in (m x m, shared) + c * S[l] + jitter * I     (S may be NULL -> in + jitter I, batch 1)
__global__ void k_big_add_diag(int m, int L, real c, real jitter, const real* __restrict__ in,
                               const real* __restrict__ S, long long s_in, real* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long mm = (long long)m * m;
    if (i >= mm * L) return;
    const long long o = i % mm, l = i / mm;
    const int r = (int)(o / m), cidx = (int)(o % m);
    out[i] = in[l * s_in + o] + (S ? c * S[i] : real(0)) + (r == cidx ? jitter : real(0));
}
// out[n][l] (+)= scale * sum_j X[l][n][j] * Kn[n][j]; one wave per (n,l)
__global__ __launch_bounds__(256) void k_big_rowdot(int b, int m, int L, real scale, const real* __restrict__ X,
                                                    long long sX, const real* __restrict__ Kn,
                                                    real* __restrict__ out, int ld_out, int col0) {
    const long long wid = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wid >= (long long)b * L) return;
    const int n = (int)(wid / L), l = (int)(wid % L);
    const real* x = X + (size_t)l * sX + (size_t)n * m;
    const real* k = Kn + (size_t)n * m;
    real s = 0;
    for (int j = lane; j < m; j += 64) s += x[j] * k[j];
    s = wave_sum(s);
    if (lane == 0) out[(size_t)n * ld_out + col0 + l] = scale * s;
}
// Two row-dot jobs over the same rows in ONE launch (blockIdx.y = job): r = k^T Si k and s = w^T Si w of the forward row stage.  Same
// arithmetic per (n, l) as two k_big_rowdot launches.
__global__ __launch_bounds__(256) void k_big_rowdot2(int b, int m, int L, real scale, const real* __restrict__ X0,
                                                     const real* __restrict__ X1, long long sX, const real* __restrict__ K0,
                                                     const real* __restrict__ K1, real* __restrict__ out0,
                                                     real* __restrict__ out1, int ld_out) {
    const long long wid = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wid >= (long long)b * L) return;
    const int n = (int)(wid / L), l = (int)(wid % L);
    const real* x = (blockIdx.y ? X1 : X0) + (size_t)l * sX + (size_t)n * m;
    const real* k = (blockIdx.y ? K1 : K0) + (size_t)n * m;
    real s = 0;
    for (int j = lane; j < m; j += 64) s += x[j] * k[j];
    s = wave_sum(s);
    if (lane == 0) (blockIdx.y ? out1 : out0)[(size_t)n * ld_out + l] = scale * s;
}
// tr(Ki A_l) = sum_ij Ki_ij A_ji and mu_l . u_l.  grid (KL_NCH, L): each workgroup sums a contiguous slice of A_l
// (coalesced; the transposed operand is the shared Ki, an L2 hit) -> part (L, KL_NCH, 2); k_big_kl adds the slices in
// index order.
#define KL_NCH 32
__global__ __launch_bounds__(256) void k_big_kl_terms(int m, const real* __restrict__ Ki, const real* __restrict__ A,
                                                      const real* __restrict__ mu, const real* __restrict__ u,
                                                      real* __restrict__ part) {
    __shared__ real red[16];
    const int l = blockIdx.y, ch = blockIdx.x;
    const long long mm = (long long)m * m, per = (mm + KL_NCH - 1) / KL_NCH, lo = ch * per, hi = lo + per < mm ? lo + per : mm;
    const real* Al = A + (size_t)l * mm;
    real tr = 0, muu = 0;
    for (long long o = lo + threadIdx.x; o < hi; o += blockDim.x) tr += Ki[o] * Al[o];      // Ki is exactly symmetric (k_symmetrize)
    if (ch == 0)
        for (int i = threadIdx.x; i < m; i += blockDim.x) muu += mu[(size_t)l * m + i] * u[(size_t)l * m + i];
    tr = block_sum(tr, red);
    muu = block_sum(muu, red);
    if (threadIdx.x == 0) { part[((size_t)l * KL_NCH + ch) * 2] = tr; part[((size_t)l * KL_NCH + ch) * 2 + 1] = muu; }
}
__global__ void k_big_kl(int m, int L, const real* __restrict__ ldK, const real* __restrict__ ldA,
                         const real* __restrict__ part, real* __restrict__ KL) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    real tr = 0;
    for (int ch = 0; ch < KL_NCH; ++ch) tr += part[((size_t)l * KL_NCH + ch) * 2];
    KL[l] = real(0.5) * (*ldK - ldA[l] - (real)m + tr + part[(size_t)l * KL_NCH * 2 + 1]);
}
// final element-wise part of the per-sample forward + partial sums
struct PostFinArgs {
    int b, L, use_rng, clip_pv;
    const real* knn; const real* q; const real* y; const real* s2; const real* eps_in; const real* state;
    real* p_m; real* p_v; real* e; real* d; real* eps; real* z; real* part;
};
__device__ __forceinline__ real philox_normal_big(unsigned long long ctr, unsigned long long idx) { return svgp_philox_normal(ctr, idx); }
// in: p_m = c k.t (done), p_v = r, e = mv, d = s  -> out: final values
__global__ __launch_bounds__(256) void k_big_post_final(PostFinArgs a) {
    __shared__ real red[16];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    real l3 = 0, ce = 0;
    if (i < a.b * a.L) {
        const int n = i / a.L;
        const real y = a.y[i], s2 = a.s2[i], p = recip_no_nan(s2), kq = a.knn[n] - a.q[n];
        const real p_m = a.p_m[i], ee = y - a.e[i], dd = kq + a.d[i] + ee * ee;
        real p_v = kq + a.p_v[i];
        if (a.clip_pv == 1) p_v = fmin(fmax(p_v, 1e-4), 100.0);
        const real ep = a.use_rng ? philox_normal_big((unsigned long long)a.state[SVGP_ST_RNG_CTR], (unsigned long long)i)
                                  : a.eps_in[i];
        a.eps[i] = ep; a.p_v[i] = p_v; a.e[i] = ee; a.d[i] = dd;
        a.z[i] = p_m + ep * sqrt(a.clip_pv == 2 ? fmin(fmax(p_v, 1e-4), 1000.0) : p_v);
        const real ls2 = log(s2), dm = p_m - y;
        l3 = real(-0.5) * (p * dd + ls2);
        ce = real(-0.5) * (real(SVGP_LOG_2PI) + ls2 + (p_v + dm * dm) * p);
    }
    l3 = block_sum(l3, red);
    ce = block_sum(ce, red);
    if (threadIdx.x == 0) { a.part[blockIdx.x * 2] = l3; a.part[blockIdx.x * 2 + 1] = ce; }
}

// ---- factor backward element-wise pieces (device scalars g3 = gT, gK) ----------------------------
struct FbArgs {
    int m, L, Ltot, geco, b_global;      // L = channels of this call (loops), Ltot = channels of the model (loss seeds)
    real c, N_train;
    const real* state;
    const real* A2;                      // Kn^T diag(g_pv_l) Kn (the reverse statistic)
    const real* SW;                      // W^T diag(p_l) W = P^T S_l P (NULL with cfg.titsias: its seed g3 is zero)
    const real* mu; const real* u; const real* ud; const real* td; const real* v;
    real* ubar; real* mubar; real* tbar;   // (L,m) each; mubar/tbar come in holding Ki ubar / K mubar
    real* Sibar; const real* Sg; const real* HG; real* Ssym;
    // rank1_late (round 6): the symmetric rank-one part (tbar v^T + v tbar^T) / 2 of X is NOT inside X -- Si X Si then lacks
    // (vbar t^T + t vbar^T) / 2 (vbar = Si tbar, t = Si v: the same terms, exactly, without two m^3 sandwiches around them), which the
    // consumers of Sg0 = -Si X Si subtract: k_big_fb_ssym here, k_big_fb_final for the channel sum.  X = A2 - g3/2 SW is then a
    // function of the statistics alone and comes out of the epilogue of the SW product (svgp_big_stats): no pass of its own.
    int rank1_late;
    const real* t; const real* vbar;
};
__device__ __forceinline__ void fb_scalars(const FbArgs& a, real& g3, real& gK) {
    const real gT = gradKL(a.geco, a.Ltot, a.state);
    g3 = svgp_seed_3(a.geco, gT); gK = svgp_seed_K(a.geco, gT, (real)a.b_global / a.N_train);
}
// (ubar = ud + gK/2 mu, mubar = Ki ubar + gK/2 u, tbar = td + c K mubar: inside k_big_gemv_fb)
// D_l = Ki - Aji_l (exactly symmetric: both operands are); the gradient of A_hat from the KL term is Abar_l = gK/2 D_l
__global__ void k_big_fb_dmat(int m, int L, const real* __restrict__ Ki, const real* __restrict__ Aji, real* __restrict__ D) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, mm = (long long)m * m;
    if (i < mm * L) D[i] = Ki[i % mm] - Aji[i];
}
// X_l = A2_l - g3/2 SW_l + (tbar_l v_l^T + v_l tbar_l^T) / 2: the gradient of Sigma_l^-1 WITHOUT the share gK/2 K D_l K that
// reaches it through A_hat = K Sigma^-1 K -- that share enters Sg = -Si Sibar Si as -gK/2 G_l D_l G_l^T (G = Si K is the forward
// product), formed early from H_l = G_l D_l, which is also Z'_l = Si K D_l: one full m^3 product per channel less than K D,
// Si (K D), (K D) K (round 4, second form).
// The gradient of t = Si v is the rank-one tbar v^T; only the symmetric part of Sg = -Si Sibar Si is ever used (Ssym = c (Sg +
// Sg^T) in the row stage, Sg + Sg^T in the kernel-matrix reverse pass, which reads Kbar_ij + Kbar_ji), so Sibar is symmetrised
// here (Sg is then symmetric up to rounding and Ssym = c (Sg + Sg^T) removes the antisymmetric residue exactly).
__global__ void k_big_fb_sibar(FbArgs a) {
    real g3, gK; fb_scalars(a, g3, gK);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, mm = (long long)a.m * a.m;
    if (i >= mm * a.L) return;
    const long long o = i % mm, l = i / mm;
    const int r = (int)(o / a.m), cidx = (int)(o % a.m);
    a.Sibar[i] = a.A2[i] - (a.SW ? real(0.5) * g3 * a.SW[i] : real(0)) +
                 real(0.5) * (a.tbar[l * a.m + r] * a.v[l * a.m + cidx] + a.tbar[l * a.m + cidx] * a.v[l * a.m + r]);
}
// The kernel below needs X and X^T of a full (non-symmetric) product.  A workgroup owns the PAIR of 32 x 32 tiles (ti, tj),
// (tj, ti), ti <= tj, of one channel: both tiles go through LDS, every global access is coalesced (the element-per-thread form
// read the transposed operand with a stride of m doubles: 0.55 + 0.61 ms per step at m = 800, L = 64).
// grid (nt, nt, L), 256 threads; thread (c = tid & 31, r0 = tid >> 5) handles rows r0 + 8 h.
// a b + c d with both products rounded on their own (no fused multiply-add: hipcc contracts by default, and HIP's __dmul_rn is a
// plain product that contracts too): the result does not depend on which product comes first
__device__ __forceinline__ real sum_of_two_products(real a, real b, real c, real d) {
#pragma clang fp contract(off)
    const real p1 = a * b, p2 = c * d;
    return p1 + p2;
}
#define TP 32
struct TilePair {
    int ti, tj, l, c, r0;
    __device__ __forceinline__ bool init() {
        ti = blockIdx.y; tj = blockIdx.x; l = blockIdx.z; c = threadIdx.x & 31; r0 = threadIdx.x >> 5;
        return ti <= tj;
    }
};
__device__ __forceinline__ void tp_load(const real* __restrict__ X, int m, const TilePair& t, real (*U)[TP + 1], real (*V)[TP + 1]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int r = t.r0 + 8 * h;
        const int ui = t.ti * TP + r, uj = t.tj * TP + t.c, vi = t.tj * TP + r, vj = t.ti * TP + t.c;
        U[r][t.c] = (ui < m && uj < m) ? X[(size_t)ui * m + uj] : real(0);
        V[r][t.c] = (vi < m && vj < m) ? X[(size_t)vi * m + vj] : real(0);
    }
    __syncthreads();
}
// Ssym = c (Sg0 + Sg0^T) - c gK HG with Sg0 = -Si X Si and HG = G D G^T (exactly symmetric: mirrored product)
__global__ __launch_bounds__(256) void k_big_fb_ssym(FbArgs a) {
    __shared__ real U[TP][TP + 1], V[TP][TP + 1];
    TilePair t;
    if (!t.init()) return;
    const int m = a.m;
    const size_t mm = (size_t)m * m, lo = (size_t)t.l * mm;
    real g3, gK; fb_scalars(a, g3, gK);
    tp_load(a.Sg + lo, m, t, U, V);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int r = t.r0 + 8 * h;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            if (side == 1 && t.ti == t.tj) break;
            const int gi = (side ? t.tj : t.ti) * TP + r, gj = (side ? t.ti : t.tj) * TP + t.c;
            if (gi < m && gj < m) {
                const size_t i = lo + (size_t)gi * m + gj;
                const real sg = side ? V[r][t.c] : U[r][t.c], sgt = side ? U[t.c][r] : V[t.c][r];
                real two = sg + sgt;
                if (a.rank1_late) {
                    const size_t vo = (size_t)t.l * m;
                    // (two separately rounded products, no fused multiply-add: element (j, i) forms the same two products in the
                    // other order, and Ssym must be symmetric bit for bit -- the packed exchange relies on it)
                    two -= sum_of_two_products(a.vbar[vo + gi], a.t[vo + gj], a.t[vo + gi], a.vbar[vo + gj]);
                }
                a.Ssym[i] = a.c * two - a.c * gK * a.HG[i];
            }
        }
    }
}
// three channel sums in ONE launch (blockIdx.y = job); same arithmetic as three k_big_sum_channels launches
__global__ void k_big_sum_channels3(int mm, int L, real scale, const real* __restrict__ in0, real* __restrict__ out0,
                                    const real* __restrict__ in1, real* __restrict__ out1, const real* __restrict__ in2,
                                    real* __restrict__ out2) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= mm) return;
    const real* in = blockIdx.y == 0 ? in0 : blockIdx.y == 1 ? in1 : in2;
    real* out = blockIdx.y == 0 ? out0 : blockIdx.y == 1 ? out1 : out2;
    real s = 0;
    for (int l = 0; l < L; ++l) s += in[(size_t)l * mm + o];
    out[o] = scale * s;
}
// out (m x m) = scale * sum over the L channel matrices
__global__ void k_big_sum_channels(int mm, int L, real scale, const real* __restrict__ in, real* __restrict__ out) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= mm) return;
    real s = 0;
    for (int l = 0; l < L; ++l) s += in[(size_t)l * mm + o];
    out[o] = scale * s;
}
// The gradient of Ki, channel sum: Kib = rep_weight (gK/2 sum_l A_l + sum_l ubar_l mu_l^T) + Qs + (Pbar K)^T
//   (tr(Ki A) and mu^T Ki mu of the KL term: window part; q_n = k^T Ki k in d and p_v and P = Ki K: rank-local row sums).
// The rank-local part is carried TRANSPOSED (the row contraction delivers Pbar^T = Wbar^T Kn): the kernel-matrix reverse pass
// reads Kbar_ij + Kbar_ji only, and Ki (A + B^T) Ki = Ki A Ki + (Ki B Ki)^T.
struct FinArgs {
    int m, L, Ltot, geco, b_global;
    real c, N_train, rep_weight;
    const real* state;
    const real* Asum; const real* ubar; const real* mu; const real* Qs; const real* PbarK;
    const real* Zs; const real* mubar; const real* t; const real* Sgs; const real* HGs; const real* Ki; const real* KiPbar;
    const real* KiKibKi;
    real* Kib; real* Kbar;
    int rank1_late; const real* vbar;      // see FbArgs: sum_l Sg0_l lacks -(vbar_l t_l^T + t_l vbar_l^T) / 2
};
__global__ void k_big_fb_kib(FinArgs a) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= a.m * a.m) return;
    const real gT = gradKL(a.geco, a.Ltot, a.state), gK = svgp_seed_K(a.geco, gT, (real)a.b_global / a.N_train);
    const int r = o / a.m, cidx = o % a.m;
    real s = real(0.5) * gK * a.Asum[o];
    for (int l = 0; l < a.L; ++l) s += a.ubar[(size_t)l * a.m + r] * a.mu[(size_t)l * a.m + cidx];
    a.Kib[o] = a.rep_weight * s + a.Qs[o] + a.PbarK[o];
}
// Kbar = rep_weight (gK/2 (Zs + Zs^T) + c sum_l mubar_l t_l^T + sum_l Sg0_l - gK/2 sum_l HG_l + gK/2 L Ki) + Ki Pbar - Ki Kib Ki
__global__ void k_big_fb_final(FinArgs a) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= a.m * a.m) return;
    const real gT = gradKL(a.geco, a.Ltot, a.state), gK = svgp_seed_K(a.geco, gT, (real)a.b_global / a.N_train);
    const int r = o / a.m, cidx = o % a.m;
    real s = real(0.5) * gK * (a.Zs[o] + a.Zs[(size_t)cidx * a.m + r] - a.HGs[o]) + a.Sgs[o] + real(0.5) * gK * (real)a.L * a.Ki[o];
    real rk = 0, rs = 0;
    for (int l = 0; l < a.L; ++l) rk += a.mubar[(size_t)l * a.m + r] * a.t[(size_t)l * a.m + cidx];
    if (a.rank1_late) {
        for (int l = 0; l < a.L; ++l)
            rs += a.vbar[(size_t)l * a.m + r] * a.t[(size_t)l * a.m + cidx] + a.t[(size_t)l * a.m + r] * a.vbar[(size_t)l * a.m + cidx];
        s -= real(0.5) * rs;
    }
    a.Kbar[o] = a.rep_weight * (s + a.c * rk) + a.KiPbar[o] - a.KiKibKi[o];
}

// ---- per-sample backward element-wise pieces ----------------------------------------------------
struct PbArgs {
    int b, m, L, geco;
    real c;
    const real* state;
    const real* y; const real* s2; const real* p_m; const real* p_v; const real* e; const real* d;
    const real* g_pv; const real* g_pm; const real* mvbar;
    const real* u; const real* t; const real* vbar;
    const real* R;      // (L,b,m) Kn Ssym_l
    const real* KnSi; long long sKS;   // forward product Kn Si_l: channel l at KnSi + l sKS (the stacked [Kn; W] Si_l block)
    const real* kSk; const real* kv;   // (b,L)
    const real* KnKi;   // (b,m)
    const real* qbar;   // (b)
    real* part;         // Knbar_part (L,b,m)
    real* Knbar; real* knnbar; real* ybar; real* s2bar;
};
// Knbar = sum_l [2 g_pv (Kn Si_l) + p (Kn Ssym_l) + mvbar u_l + c g_pm t_l + p y vbar_l] + 2 qbar (Kn Ki); knnbar = -qbar.
// Kn Si is the forward pass's product, R = Kn Ssym (the d-term's k^T Ki A Ki k reaches Kn through Wbar (Ki K)^T, one product for all
// channels: svgp_big_posterior_bwd).  One pass over the two (L, b, m) arrays with the channel sum in registers (in channel order, as
// the former pair k_big_pb_part -> (L, b, m) partials -> k_big_pb_sum added them: 2 x 205 MB less traffic at the SPRITES shape).
// Workgroup = one batch row n (its b x L scalars staged in LDS once), threads stride the columns.
__global__ __launch_bounds__(256) void k_big_pb_knbar(PbArgs a) {
    extern __shared__ real pb_lds[];
    const int n = blockIdx.x, L = a.L, m = a.m;
    real* c0 = pb_lds;            // 2 g_pv
    real* c1 = c0 + L;            // p
    real* c2 = c1 + L;            // mvbar
    real* c3 = c2 + L;            // c g_pm
    real* c4 = c3 + L;            // p y
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        const size_t e = (size_t)n * L + l;
        const real p = recip_no_nan(a.s2[e]);
        c0[l] = real(2) * a.g_pv[e]; c1[l] = p; c2[l] = a.mvbar[e]; c3[l] = a.c * a.g_pm[e]; c4[l] = p * a.y[e];
    }
    __syncthreads();
    const long long bm = (long long)a.b * m;
    const real qbar = a.qbar[n];
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        const size_t o = (size_t)n * m + j;
        real acc = 0;
#pragma unroll 4
        for (int l = 0; l < L; ++l) {
            const size_t vi = (size_t)l * m + j;
            acc += c0[l] * a.KnSi[(size_t)l * a.sKS + o] + c1[l] * a.R[(size_t)l * bm + o] + c2[l] * a.u[vi] + c3[l] * a.t[vi] +
                   c4[l] * a.vbar[vi];
        }
        a.Knbar[o] = acc + real(2) * qbar * a.KnKi[o];
    }
    if (threadIdx.x == 0) a.knnbar[n] = -qbar;
}
__global__ void k_big_pb_elem(PbArgs a) {     // ybar, s2bar
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.b * a.L) return;
    const real gT = gradKL(a.geco, a.L, a.state), g3 = svgp_seed_3(a.geco, gT);
    const real y = a.y[i], s2 = a.s2[i], p = recip_no_nan(s2), dm = a.p_m[i] - y, kV = a.kv[i];
    const real pbar = real(-0.5) * g3 * a.d[i] + a.kSk[i] + y * kV;
    a.ybar[i] = -gT * p * dm - g3 * p * a.e[i] + p * kV;
    a.s2bar[i] = real(0.5) * gT * (p - (a.p_v[i] + dm * dm) * p * p) - real(0.5) * g3 * p - pbar * p * p;
}
// y_l = alpha A_l x_l for L channels (A_l: m x m row-major at A + l sA, sA = 0: one shared matrix; x, y: (L, m)).
// grid (ceil(m / 16), L), 256 threads: wave w takes rows 4 w .. 4 w + 3 of the block, lanes stride the row, x_l sits in LDS.
// The batched GEMM spends a 32- or 64-wide tile on the single column (m = 800, L = 64: 130 us; this: one pass over A).
__global__ __launch_bounds__(256) void k_big_gemv(int m, real alpha, const real* __restrict__ A, long long sA,
                                                  const real* __restrict__ x, real* __restrict__ y) {
    extern __shared__ real xs[];
    const int l = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const real* xl = x + (size_t)l * m;
    for (int k = threadIdx.x; k < m; k += 256) xs[k] = xl[k];
    __syncthreads();
    const int i0 = blockIdx.x * 16 + w * 4;
    const real* Al = A + (size_t)l * sA;
    real acc[4] = {0, 0, 0, 0};
    for (int k = lane; k < m; k += 64) {
        const real xv = xs[k];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (i0 + r < m) acc[r] += Al[(size_t)(i0 + r) * m + k] * xv;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        real v = acc[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0 && i0 + r < m) y[(size_t)l * m + i0 + r] = alpha * v;
    }
}

// The vector chain of the late reverse factor stage with its element-wise steps inside the two matrix-vector products (round 5:
// ubar -> Ki ubar -> mubar -> K mubar -> tbar was three element kernels + two products = five dependent launches of 4-8 us):
//   MODE 1: x = ubar = ud + gK/2 mu (stored by the first row block), y = Ki x, result mubar = y + gK/2 u;
//   MODE 2: x = mubar,                                               y = K x,  result tbar  = td + c y.
// Same grid and summation order as k_big_gemv; the element-wise expressions are those of the three kernels they replace.
template <int MODE>
__global__ __launch_bounds__(256) void k_big_gemv_fb(FbArgs a, const real* __restrict__ A) {
    extern __shared__ real xs[];
    const int m = a.m, l = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    real g3, gK; fb_scalars(a, g3, gK);
    for (int k = threadIdx.x; k < m; k += 256) {
        const size_t i = (size_t)l * m + k;
        real xv;
        if (MODE == 1) {
            xv = a.ud[i] + real(0.5) * gK * a.mu[i];
            if (blockIdx.x == 0) a.ubar[i] = xv;
        } else {
            xv = a.mubar[i];
        }
        xs[k] = xv;
    }
    __syncthreads();
    const int i0 = blockIdx.x * 16 + w * 4;
    real acc[4] = {0, 0, 0, 0};
    for (int k = lane; k < m; k += 64) {
        const real xv = xs[k];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (i0 + r < m) acc[r] += A[(size_t)(i0 + r) * m + k] * xv;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        real v = acc[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0 && i0 + r < m) {
            const size_t i = (size_t)l * m + i0 + r;
            if (MODE == 1) {
                real y = real(1) * v;
                y += real(0.5) * gK * a.u[i];
                a.mubar[i] = y;
            } else {
                const real y = real(1) * v;
                a.tbar[i] = a.td[i] + a.c * y;
            }
        }
    }
}

inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }
inline bool x0_epilogue_on() {
    const char* e = getenv("SVGP_X0_EPILOGUE");
    return !(e && e[0] == '0');
}

}  // namespace

#define RUNC(call) do { int rc__ = (call); if (rc__) return rc__; } while (0)
// A symmetric right operand (K_mm, the inverses Ki / Sigma^-1 / (A_hat + jI)^-1 -- made exactly symmetric by the inverse,
// linalg.hip k_symmetrize -- and the mirrored-store products A_hat, M2, Qm, Ssym, Abar) is passed as stored-transposed
// (tb = 1): the same product bit for bit, and the [j][k] staging form of the B operand runs 46 against 42 TFLOP/s
// for [k][j] at 800^3 x 64.
// cfg.gemm_f32 = 1: every product on the float32 MFMA (float64 storage); GEMM_S: the statistics products, also with 2
#define GEMM(...) RUNC((c->gemm_f32 == 1 ? svgp_dgemm_f32c_batched : svgp_dgemm_batched)(__VA_ARGS__, stream))
#define GEMM_S(...) RUNC((c->gemm_f32 ? svgp_dgemm_f32c_batched : svgp_dgemm_batched)(__VA_ARGS__, stream))
// products whose result is symmetric (K Si K, Ki A Ki, Ki S Ki, K Abar K, Kn^T diag(w) Kn): lower tiles + mirrored stores.
// arguments: ta, tb, M (= N), K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, batch
#define GEMV(alpha_, A_, sA_, x_, y_, L_)                                                                                \
    do {                                                                                                                 \
        hipLaunchKernelGGL(k_big_gemv, dim3((m + 15) / 16, (L_)), dim3(256), (size_t)m * sizeof(real), (hipStream_t)stream, m,  \
                           real(alpha_), A_, (long long)(sA_), x_, y_);                                                   \
        SVGP_LAUNCH_CHECK();                                                                                             \
    } while (0)
#define GEMM_SYM(...) RUNC(svgp_dgemm_symout_batched(c->gemm_f32 == 1, __VA_ARGS__, stream))
#define GEMM_S_SYM(...) RUNC(svgp_dgemm_symout_batched(c->gemm_f32 != 0, __VA_ARGS__, stream))

// scratch carving (layout fields scr_bm / scr_mm / scr_vec / scr_inv / scr_bl / scr_sm are allocated by api.hip)
struct BigScr {
    real *bm, *bm2, *mm0, *mm1, *mm2, *mm3, *vec0, *vec1, *vec2, *trm, *ldtmp, *qbar, *inv, *bl0, *bl1, *wst;
    // forward products kept for the reverse pass (behind the (L, b, m) scratch in scr_bm): KS = [Kn; W] Si_l (L, 2b, m) --
    // channel l at KS + l sKS, Kn Si_l first, W Si_l at + b m --, Kn Ki (b, m); Wbar (b, m)
    real *KS, *KnKi, *X, *Wbar, *W, *sk2;      // X (b, 2m) = [qbar * Kn | Wbar]: Wbar = X + m, leading dimension 2 m
    long long sKS, sk2_elems;
    // channel-independent m x m matrices: P^T = K Ki, Pbar = Kn^T Wbar, Qs = Kn^T diag(qbar) Kn, sum_l A_l, sum_l Z'_l, sum_l Sg_l,
    // two temporaries
    real *PT, *Pbar, *Qs, *Asum, *Zs, *Sgs, *tA, *tB, *tC, *HGs;
};
static BigScr big_scr(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, double* ws) {
    const size_t Lmm = (size_t)c->L * c->m * c->m, Lm = (size_t)c->L * c->m, bL = (size_t)c->b * c->L, mm = (size_t)c->m * c->m;
    const size_t bcap = (size_t)(c->b_cap > 0 ? c->b_cap : c->b);
    BigScr s;
    s.bm = ws + wl.scr_bm; s.bm2 = ws + wl.Knbar_part;
    const size_t Lbm_cap = (size_t)c->L * bcap * c->m;      // the layout is sized for the capacity
    s.KS = s.bm + Lbm_cap; s.sKS = 2LL * c->b * c->m; s.KnKi = s.KS + 2 * Lbm_cap; s.X = s.KnKi + bcap * c->m; s.Wbar = s.X + c->m;
    s.W = ws + wl.Kn + (size_t)c->b * c->m;
    s.mm0 = ws + wl.scr_mm; s.mm1 = s.mm0 + Lmm; s.mm2 = s.mm1 + Lmm; s.mm3 = s.mm2 + Lmm;
    s.vec0 = ws + wl.scr_vec; s.vec1 = s.vec0 + Lm; s.vec2 = s.vec1 + Lm; s.trm = s.vec2 + Lm; s.ldtmp = s.trm + 2 * c->L;
    s.qbar = s.ldtmp + c->L + 16;
    s.inv = ws + wl.scr_inv;
    s.bl0 = ws + wl.scr_bl; s.bl1 = s.bl0 + bL; s.wst = s.bl0;      // p = 1 / s2 (b, L): weights of the early SW statistic
    real* sm = ws + wl.scr_sm;
    s.PT = sm; s.Qs = sm + mm; s.Pbar = sm + 2 * mm;       // Qs | Pbar^T contiguous: one (2 m, m) product writes both
     s.Asum = sm + 3 * mm; s.Zs = sm + 4 * mm; s.Sgs = sm + 5 * mm;
    s.tA = sm + 6 * mm; s.tB = sm + 7 * mm; s.tC = sm + 8 * mm; s.HGs = sm + 9 * mm;
    s.sk2 = sm + 10 * mm; s.sk2_elems = svgp_dgemm_splitk_scratch_elems(2 * c->m, c->m, (int)bcap);
    return s;
}

int svgp_big_stats(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, double* ws, const double* state, int mode,
                   void* stream) {
    const int b = c->b, m = c->m, L = c->L;
    const real cc = c->N_train / (double)c->b_global;
    const BigScr s = big_scr(c, wl, ws);
    hipStream_t st = (hipStream_t)stream;
    real* Kn = ws + wl.Kn;
    // weights: forward uses (g_pv, g_pm) as temporaries for (p, p*y); backward fills g_pv, mvbar, g_pm
    real* wbuf = ws + wl.g_pv;
    real* abuf = mode == 0 ? ws + wl.g_pm : ws + wl.mvbar;
    real* bbuf = ws + wl.g_pm;
    // reverse statistics: ud = mvbar^T Kn and td = c g_pm^T Kn as ONE split-K contraction over the rows, operands stacked (b, 2 L) in
    // scr_bl (free until the row-form SW weights / the reverse row stage), results in the contiguous [ud | td] of statB, the factor c
    // applied to td's rows by the reduction pass: the same partial sums per element as two contractions, two launches less
    const bool stacked = mode == 1 && wl.td == wl.ud + (int64_t)L * m;
    hipLaunchKernelGGL(k_big_weights, dim3(nblk((long long)b * L)), dim3(256), 0, st, b * L, L, mode, SVGP_LOSS_FLAGS(c), c->clip_pv, cc, state,
                       ws + wl.qnet_mu, ws + wl.qnet_var, ws + wl.p_m, ws + wl.p_v, ws + wl.e, ws + wl.eps,
                       ws + wl.zbar, wbuf, abuf, bbuf, stacked ? s.bl0 : (real*)nullptr);
    SVGP_LAUNCH_CHECK();
    real* S = mode == 0 ? ws + wl.S : ws + wl.A2;
    real* v1 = mode == 0 ? ws + wl.v : ws + wl.ud;
    // S_l = Kn^T diag(w_l) Kn: the weights w[n][l] scale the rows of the B operand while they are staged (no (L, b, m) copy
    // of the scaled K_nm: config 3 saved a 15 us launch and 67 MB of traffic per statistics stage).
    RUNC(svgp_dgemm_symout_batched(c->gemm_f32 != 0, 1, 0, m, b, 1.0, Kn, m, 0, Kn, m, 0, 0.0, S, m, (long long)m * m, L, stream, wbuf, L,
                                   1));
    // v1 (L x m) = a^T Kn: L x m outputs and a contraction over the batch -> split-K (8 tiles of 32 otherwise walk all b rows)
    const long long sk = svgp_dgemm_splitk_scratch_elems(L, m, b);
    SVGP_REQUIRE(sk >= 0 && sk <= (long long)c->L * m * m, SVGP_ERR_INVALID, "split-K scratch");
    // scratch: the second half of fb_part (unused since round 4; scr_mm may be in use by the early reverse half on the side stream
    // while the reverse statistics run)
    real* sks = ws + wl.fb_part + (size_t)c->L * m * m;
    if (stacked) {
        const long long sk2 = svgp_dgemm_splitk_scratch_elems(2 * L, m, b);
        SVGP_REQUIRE(sk2 >= 0 && sk2 <= (long long)c->L * m * m, SVGP_ERR_INVALID, "split-K scratch");
        RUNC(svgp_dgemm_splitk_rows2(1, 0, 2 * L, m, b, 1.0, cc, L, s.bl0, 2 * L, Kn, m, 0.0, v1, m, sks, sk2, stream));
    } else {
        RUNC(svgp_dgemm_splitk(1, 0, L, m, b, 1.0, abuf, L, Kn, m, 0.0, v1, m, sks, sk, stream));
    }
    if (mode == 1) {
        if (!stacked) RUNC(svgp_dgemm_splitk(1, 0, L, m, b, cc, bbuf, L, Kn, m, 0.0, ws + wl.td, m, sks, sk, stream));
        // the rank-local row sums of the reverse pass (they enter Kbar linearly: no exchange, see the file header):
        // [Qs; Pbar^T] (2 m, m) = X^T Kn, split over the rows when that leaves few output tiles (config 3: 23.6 + 25.9 us for the two
        // products as single launches with a contraction of 1024, round 4)
        hipLaunchKernelGGL(k_big_rowlocal, dim3(nblk((long long)b * m)), dim3(256), 0, st, b, m, L, SVGP_LOSS_FLAGS(c), state,
                           ws + wl.qnet_var, wbuf, Kn, s.KS + (size_t)b * m, s.sKS, s.X, s.qbar);
        SVGP_LAUNCH_CHECK();
        RUNC(svgp_dgemm_splitk(1, 0, 2 * m, m, b, 1.0, s.X, 2 * m, Kn, m, 0.0, s.Qs, m, s.sk2, s.sk2_elems, stream));   // Qs | Pbar^T (contiguous)
        // SW_l = W^T diag(p_l) W in its row form (all rows local, b < 3 m; svgp_big_factor_bwd has the rule): here, at the end of the
        // reverse statistics on the caller's stream, which then waits for the side branch anyway.  (SPRITES m = 800, kernel trace of
        // round 4: issued early on a third stream it ran 1.4 ms instead of 0.35 beside the row stage's product and the tail's
        // factorisation; A / B in one run: 18.15 vs 18.20 ms per step, i.e. no difference -- this form needs one stream less.)
        // mm2 and the weights in scr_bl are untouched by the side branch.
        // Round 6: the product's epilogue writes X0 = A2 - g3/2 SW in SW's place (mm2) -- what the late reverse factor half needs
        // of SW; the rank-one part of X is applied after the Sigma^-1 sandwiches (FbArgs.rank1_late), so the 1 GB pass k_big_fb_sibar
        // (0.42 ms at m = 800, L = 64) is gone.  -g3/2 is a device scalar (the loss seeds live in the state vector): written by
        // k_big_recip, read by the epilogue.  SVGP_X0_EPILOGUE=0: plain SW + the pass (A / B measurements).
        if (!c->titsias && c->b == c->b_global && c->b < 3 * m) {
            real* mhalf_g3 = s.ldtmp + c->L + 8;       // (slots L + 1 .. L + 15 of the log-det scratch are free)
            hipLaunchKernelGGL(k_big_recip, dim3(nblk((long long)b * L)), dim3(256), 0, st, b * L, ws + wl.qnet_var, s.wst,
                               SVGP_LOSS_FLAGS(c), L, state, mhalf_g3);
            SVGP_LAUNCH_CHECK();
            svgp_gemm_epi ep;
            ep.E = ws + wl.A2; ep.lde = m; ep.se = (long long)m * m; ep.g1 = 1.0; ep.alpha_dev = mhalf_g3;
            ep.e_sym = 1;                              // A2 = Kn^T diag(g_pv) Kn: a mirrored-store product
            RUNC(svgp_dgemm_symout_batched(c->gemm_f32 != 0, 1, 0, m, b, 1.0, s.W, m, 0, s.W, m, 0, 0.0, s.mm2, m, (long long)m * m, L,
                                           stream, s.wst, L, 1, x0_epilogue_on() ? &ep : nullptr));
        }
    }
    // (K_mm + jI)^-1 and its log det (SVGPVAE_model.py:239,270,273) are formed by svgp_big_factor_fwd, in the same
    // launches as the L channel inverses
    return SVGP_OK;
}

// Channel window [l0, l0 + nl): the factor stage of those channels only (all of them: l0 = 0, nl = L).  With the batch
// sharded over ranks and the statistics reduce-SCATTERED over channels, every rank factors L / G channels instead of all
// L redundantly; (K_mm + jI)^-1, q_n and W are channel-independent and computed by every caller.
// part: 0 = the whole stage; 1 = without its tail -- (A_hat_l + jI)^-1, its log det and the KL_l scalars, which only the
// reverse factor stage and the final ELBO need; 2 = that tail alone.  The training step issues the tail on a side stream
// (api.hip, sprites.py) so that the second batched inverse of the step runs beside the row stage, the decoder and the
// reverse statistics instead of in front of them.  The tail touches A (read), Aji, KL, the inverse workspace, s.ldtmp and
// the trace partials in fb_part -- nothing the stages between the two factor stages use.
int svgp_big_factor_fwd(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, double* ws, void* stream, int l0, int nl,
                        int part) {
    const int b = c->b, m = c->m, L = nl;
    const long long mm = (long long)m * m;
    const real cc = c->N_train / (double)c->b_global;
    const BigScr s = big_scr(c, wl, ws);
    hipStream_t st = (hipStream_t)stream;
    const size_t om = (size_t)l0 * mm, ov = (size_t)l0 * m;
    real *K = ws + wl.K, *Ki = ws + wl.Ki, *Si = ws + wl.Si + om, *G = ws + wl.G + om, *A = ws + wl.A + om, *Aji = ws + wl.Aji + om;
    real *t = ws + wl.t + ov, *mu = ws + wl.mu_hat + ov, *u = ws + wl.u + ov, *v = ws + wl.v + ov, *Kn = ws + wl.Kn;
    real* klp = ws + wl.fb_part;                 // (L, KL_NCH, 2) trace partials (fb_part is free until the reverse factor stage)
    if (part == 2) goto aji_tail;
    {
    // part 5 = the CHANNEL-INDEPENDENT block alone -- (K + jI)^-1 and its log det, Kn Ki, q, W, P^T: functions of the kernel
    // matrices only, not of the encoder's output -- which the training step issues on a side branch as soon as the kernel
    // matrices exist, beside the encoder's tail, the forward statistics and the channel inverses (api.hip, comm.hip); part 6 = the
    // channel block up to mu (Sigma^-1, t, G, A_hat); part 7 = what needs both (u = Ki mu, the KL terms).  1 = 5 + 6 + 7 in one
    // call, with the (K + jI) inverse riding in the channel matrices' launches.  Same operations on the same values either way.
    const bool split = part == 5 || part == 6 || part == 7;
    SVGP_REQUIRE(!split || m < SVGP_CHOL_INVERSE_MIN_M, SVGP_ERR_INVALID, "the split forward factor stage exists for m < %d",
                 SVGP_CHOL_INVERSE_MIN_M);
    const bool do_k = !split || part == 5, do_sig = !split || part == 6, do_kl = !split || part == 7;
    // the blocked inverse's workspace is (pivots | ping-pong copy) per matrix: the channel batch takes the head, (K + jI) the tail
    real* inv_k = s.inv + (size_t)c->L * (2 * 32 * 32 + (size_t)mm);
    if (do_sig) {
        hipLaunchKernelGGL(k_big_add_diag, dim3(nblk(mm * L)), dim3(256), 0, st, m, L, cc, c->jitter, K, ws + wl.S + om, 0LL, Si);
        SVGP_LAUNCH_CHECK();
    }
    if (do_k) {
        // K_mm inverse + log det (SVGPVAE_model.py:239,270,273) next to the channel matrices (:331)
        hipLaunchKernelGGL(k_big_add_diag, dim3(nblk(mm)), dim3(256), 0, st, m, 1, real(0), c->jitter, K, (const real*)nullptr,
                           0LL, Ki);
        SVGP_LAUNCH_CHECK();
    }
    if (split) {
        if (part == 5) RUNC(svgp_spd_inverse_fused(m, 1, Ki, ws + wl.ldK, 0, nullptr, nullptr, inv_k, stream));
        if (part == 6) RUNC(svgp_spd_inverse_fused(m, L, Si, s.ldtmp, 0, nullptr, nullptr, s.inv, stream));
    } else if (m < SVGP_CHOL_INVERSE_MIN_M) {
        RUNC(svgp_spd_inverse_fused(m, L, Si, s.ldtmp, 1, Ki, ws + wl.ldK, s.inv, stream));
    } else if (Ki == Si + (size_t)L * mm) {
        // one batch of L + 1: Ki sits right behind Si in the workspace (api.hip); its log det is the last entry
        RUNC(svgp_spd_inverse_batched(m, L + 1, Si, s.ldtmp, s.inv, stream));
        SVGP_CHECK_HIP(hipMemcpyAsync(ws + wl.ldK, s.ldtmp + L, sizeof(real), hipMemcpyDeviceToDevice, st));
    } else {
        RUNC(svgp_spd_inverse_batched(m, 1, Ki, ws + wl.ldK, s.inv, stream));
        RUNC(svgp_spd_inverse_batched(m, L, Si, s.ldtmp, s.inv, stream));
    }
    if (do_sig) {
        GEMV(1.0, Si, mm, v, t, L);                                                                  // t = Si v
        GEMM(0, 1, m, m, m, 1.0, Si, m, mm, K, m, 0, 0.0, G, m, mm, L);                              // G = Si K   (K = K^T read as [j][k])
        {   // A = K G = K Si K, and A + jI beside it as a second output (the input of the tail's inverse).  A / B in one run at m = 800:
            // 18.15 ms per step against 18.26 with a copy pass in front of the tail's inverse -- the product gets 240 us slower (four
            // stores per element, two of them strided mirror stores), the side branch, which is the longer one, 100-150 us shorter
            svgp_gemm_epi ep;
            ep.C2 = Aji; ep.sc2 = mm; ep.a2 = 1.0; ep.d2 = c->jitter;
            RUNC(svgp_dgemm_symout_batched(c->gemm_f32 == 1, 0, 0, m, m, 1.0, K, m, 0, G, m, mm, 0.0, A, m, mm, L, stream, nullptr, 0, 0, &ep));
        }
        GEMV(cc, K, 0, t, mu, L);                                                                    // mu = c K t
    }
    if (do_kl) {
        GEMV(1.0, Ki, 0, mu, u, L);                                                                  // u = Ki mu
        hipLaunchKernelGGL(k_big_kl_terms, dim3(KL_NCH, L), dim3(256), 0, st, m, Ki, A, mu, u, klp);
        SVGP_LAUNCH_CHECK();
    }
    if (do_k) {
        // q_n = k_n^T Ki k_n;  W = (Kn Ki) K behind the rows of Kn;  P^T = K Ki
        GEMM(0, 1, b, m, m, 1.0, Kn, m, 0, Ki, m, 0, 0.0, s.KnKi, m, 0, 1);           // kept: the reverse pass reads Kn Ki again
        hipLaunchKernelGGL(k_big_rowdot, dim3(nblk((long long)b * 64)), dim3(256), 0, st, b, m, 1, real(1), s.KnKi, 0LL, Kn,
                           ws + wl.q, 1, 0);
        SVGP_LAUNCH_CHECK();
        GEMM(0, 1, b, m, m, 1.0, s.KnKi, m, 0, K, m, 0, 0.0, s.W, m, 0, 1);
        GEMM(0, 1, m, m, m, 1.0, K, m, 0, Ki, m, 0, 0.0, s.PT, m, 0, 1);
    }
    if (part == 1 || split) return SVGP_OK;
    }
aji_tail:
    // (Aji holds A_hat + jI: written by the product A = K G above; the tail inverts it in place)
    RUNC(svgp_spd_inverse_batched(m, L, Aji, s.ldtmp, s.inv, stream));
    hipLaunchKernelGGL(k_big_kl, dim3(nblk(L)), dim3(256), 0, st, m, L, ws + wl.ldK, s.ldtmp, klp, ws + wl.KL + l0);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

int svgp_big_posterior_fwd(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, const double* eps, double* ws,
                           double* state, void* stream) {
    const int b = c->b, m = c->m, L = c->L;
    const long long mm = (long long)m * m, bm = (long long)b * m;
    const real cc = c->N_train / (double)c->b_global;
    const BigScr s = big_scr(c, wl, ws);
    hipStream_t st = (hipStream_t)stream;
    real* Kn = ws + wl.Kn;
    // [Kn; W] Si_l in one product over the 2 b stacked rows; kept for svgp_big_posterior_bwd / svgp_big_stats (mode 1)
    GEMM(0, 1, 2 * b, m, m, 1.0, Kn, m, 0, ws + wl.Si, m, mm, 0.0, s.KS, m, s.sKS, L);
    // r = k^T Si k -> p_v slot and s = w^T Si w (= k^T Ki A Ki k) -> d slot: one launch, two jobs
    hipLaunchKernelGGL(k_big_rowdot2, dim3(nblk((long long)b * L * 64), 2), dim3(256), 0, st, b, m, L, real(1), s.KS, s.KS + bm, s.sKS,
                       Kn, (const real*)s.W, ws + wl.p_v, ws + wl.d, L);
    SVGP_LAUNCH_CHECK();
    GEMM(0, 1, b, L, m, cc, Kn, m, 0, ws + wl.t, m, 0, 0.0, ws + wl.p_m, L, 0, 1);               // p_m = c Kn t^T
    GEMM(0, 1, b, L, m, 1.0, Kn, m, 0, ws + wl.u, m, 0, 0.0, ws + wl.e, L, 0, 1);                // mv -> e slot
    PostFinArgs a;
    a.b = b; a.L = L; a.use_rng = eps == nullptr; a.clip_pv = c->clip_pv;
    a.knn = ws + wl.knn; a.q = ws + wl.q; a.y = ws + wl.qnet_mu; a.s2 = ws + wl.qnet_var; a.eps_in = eps; a.state = state;
    a.p_m = ws + wl.p_m; a.p_v = ws + wl.p_v; a.e = ws + wl.e; a.d = ws + wl.d; a.eps = ws + wl.eps; a.z = ws + wl.z;
    a.part = ws + wl.part_sums + (size_t)svgp_n_part(c) * 4;
    const unsigned nb = nblk((long long)b * L);
    SVGP_REQUIRE((long long)nb <= wl.n_post, SVGP_ERR_INVALID, "partial-sum layout too small");
    hipLaunchKernelGGL(k_big_post_final, dim3(nb), dim3(256), 0, st, a);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// channel window [l0, l0 + nl) as in svgp_big_factor_fwd; Kbar then holds rep_weight x the window's share of the gradient of
// K_mm plus this rank's row-local share (the shares of the ranks add up in the gradient exchange: kernel_matrix_bwd is linear
// in Kbar and takes it unweighted on this path).
// part: 0 = the whole stage; 1 = its EARLY half (3 + 4: its two parts separately); 2 = its LATE half (6 + 7: see there).  The early half --
//   D = Ki - Aji, H = G D (= Z' = Sigma^-1 K D), HG = H G^T (= Sigma^-1 K D K Sigma^-1), the channel sums of H, HG and A_hat
// -- depends on forward quantities only (the scalar gK/2 of Abar = gK/2 D is applied where the products are consumed),
// not on the reverse statistics B2, ud, td.  The training step issues it on the side stream right behind the forward stage's
// tail, under the row stage, the networks and the reverse statistics; the late half (ubar ... Sigma^-1 Sibar Sigma^-1, the
// Ki-gradient, Kbar) stays on the critical path.  Same operations on the same values either way.  Buffers: H in mm0, HG in mm3, then X in mm1, Sigma^-1 X in mm0 and Sg0 in mm1; SW in mm2 (its m-space form: T in mm1, early).
// Part 3 (needs no (A_hat + jI)^-1: can run beside the forward tail) is the statistic SW (see there).
int svgp_big_factor_bwd(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, double* ws, const double* state,
                        void* stream, int l0, int nl, int part) {
    const int m = c->m, L = nl;
    const long long mm = (long long)m * m;
    const real cc = c->N_train / (double)c->b_global;
    const BigScr s = big_scr(c, wl, ws);
    hipStream_t st = (hipStream_t)stream;
    const size_t om = (size_t)l0 * mm, ov = (size_t)l0 * m;
    real *K = ws + wl.K, *Ki = ws + wl.Ki, *Si = ws + wl.Si + om, *A = ws + wl.A + om, *Aji = ws + wl.Aji + om;
    FbArgs a;
    a.m = m; a.L = L; a.Ltot = c->L; a.geco = SVGP_LOSS_FLAGS(c); a.b_global = c->b_global; a.c = cc; a.N_train = c->N_train; a.state = state;
    a.A2 = ws + wl.A2 + om; a.mu = ws + wl.mu_hat + ov; a.u = ws + wl.u + ov; a.ud = ws + wl.ud + ov; a.td = ws + wl.td + ov;
    a.v = ws + wl.v + ov;
    a.ubar = s.vec0; a.mubar = s.vec1; a.tbar = s.vec2; a.Sibar = s.mm1; a.Sg = s.mm1; a.HG = s.mm3; a.Ssym = ws + wl.Ssym + om;
    const unsigned gmm = nblk(mm * L), ntp = (unsigned)((m + TP - 1) / TP);
    // SW_l = W^T diag(p_l) W = P^T S_l P -- forward quantities only.  Over the rows (a statistics product with contraction b) when ALL
    // rows of the batch are local (b == b_global) and that is the cheaper form (m^2 b against 3 m^3 per channel): at the end of the
    // reverse statistics (svgp_big_stats, mode 1).  From S_l otherwise -- under data parallelism S_l is the all-reduced statistic, so SW needs no exchange of
    // its own (3 m^3 L / G flops per rank on the window): early half, first part (3), which needs not even (A_hat + jI)^-1 and can
    // run beside the forward tail.  mm2; T = S P in mm1 (free until the late half).
    const bool has_sw = !c->titsias;
    a.SW = has_sw ? s.mm2 : nullptr;
    const bool sw_rows = c->b == c->b_global && c->b < 3 * m;
    // X0 = A2 - g3/2 SW sits in mm2 in SW's place (epilogue of the SW product, svgp_big_stats mode 1; all L channels: the row form
    // exists for b == b_global, i.e. without a channel window).  mm2 is only read here: the stage can be repeated on a workspace.
    const bool x0_ready = has_sw && sw_rows && l0 == 0 && nl == c->L && x0_epilogue_on();
    a.rank1_late = x0_ready ? 1 : 0; a.t = ws + wl.t + ov; a.vbar = ws + wl.vbar + ov;
    if (has_sw && !sw_rows && (part == 0 || part == 1 || part == 3)) {
        GEMM(0, 1, m, m, m, 1.0, ws + wl.S + om, m, mm, s.PT, m, 0, 0.0, s.mm1, m, mm, L);        // T = S P   (P = (P^T)^T)
        GEMM_SYM(0, 0, m, m, 1.0, s.PT, m, 0, s.mm1, m, mm, 0.0, s.mm2, m, mm, L);                // SW = P^T T
    }
    if (part == 3) return SVGP_OK;
    if (part == 0 || part == 1 || part == 4) {
        real* G = ws + wl.G + om;
        // H = G D = Si K (Ki - Aji) = Z', D = D^T read as [j][k].  Small m (launch-bound, config 3): D = Ki - Aji is formed while the
        // B operand is staged (one launch less: 1.307 -> 1.299 ms).  Large m: a pass materialises D first -- the second operand
        // stream inside the main loop costs the product more than the pass (measured at m = 800: 18.15 -> 18.7 ms per step).
        if (m <= 512) {
            RUNC(svgp_dgemm_bsub_batched(c->gemm_f32 == 1, 0, 1, m, m, m, 1.0, G, m, mm, Aji, m, mm, Ki, 0.0, s.mm0, m, mm, L, stream));
        } else {
            real* Db = ws + wl.fb_part;      // (first half of fb_part: the forward tail's trace partials at its head are consumed)
            hipLaunchKernelGGL(k_big_fb_dmat, dim3(gmm), dim3(256), 0, st, m, L, Ki, Aji, Db);
            SVGP_LAUNCH_CHECK();
            GEMM(0, 1, m, m, m, 1.0, G, m, mm, Db, m, mm, 0.0, s.mm0, m, mm, L);
        }
        GEMM_SYM(0, 1, m, m, 1.0, s.mm0, m, mm, G, m, mm, 0.0, s.mm3, m, mm, L);         // HG = H G^T = Si K D K Si  (mm3)
        hipLaunchKernelGGL(k_big_sum_channels3, dim3(nblk(mm), 3), dim3(256), 0, st, (int)mm, L, real(1), (const real*)s.mm0, s.Zs,
                           (const real*)s.mm3, s.HGs, (const real*)A, s.Asum);
        SVGP_LAUNCH_CHECK();
    }
    if (part == 1 || part == 4) return SVGP_OK;
    // Round 6: part 7 in three pieces -- 8 = the channel block (X sandwiches if they are not part 6's, Ssym, the channel sum Sgs), 9 = the
    // single-matrix chain of the gradient of Ki (K Pbar^T, Kib, Ki Kib Ki, Pbar^T Ki: five small launches that read nothing of 8) and
    // 10 = the closing assembly of Kbar.  7 = 8 + 9 + 10; a caller with a free side branch runs 9 beside 8.
    const bool run_8 = part != 9 && part != 10, run_9 = part != 8 && part != 10, run_10 = part != 8 && part != 9;
    if (run_8) {
    // The late half in two parts (round 5): 6 = what reads NOTHING the early half writes, 7 = the rest (2 = 6 + 7).  The vector chain
    // always belongs to part 6; X, vbar and the two full products Si X, -(Si X) Si do when the statistic SW comes from the reverse
    // statistics on the caller's stream (`sw_rows`: SPRITES) and not from the early half -- then Si X goes to the Ssym slot (free
    // until the tile-pair kernel below writes it) instead of mm0, where the early half keeps H until it is summed.  A caller with
    // the early half on a side branch issues part 6, THEN joins, then part 7: at m = 800 the caller's stream waited 0.8 ms at the join
    // with 2.3 ms of its own work ready (kernel trace).
    const bool do_a = part != 7 && part != 8, do_b = part != 6, x_early = sw_rows || !has_sw;
    real* six = x_early ? ws + wl.Ssym + om : s.mm0;
    auto x_block = [&]() -> int {
        if (!x0_ready) {
            hipLaunchKernelGGL(k_big_fb_sibar, dim3(gmm), dim3(256), 0, st, a);            // X (mm1)
            SVGP_LAUNCH_CHECK();
        }
        GEMV(1.0, Si, mm, s.vec2, ws + wl.vbar + ov, L);                                   // vbar = Si tbar
        GEMM(0, 0, m, m, m, 1.0, Si, m, mm, x0_ready ? s.mm2 : s.mm1, m, mm, 0.0, six, m, mm, L);   // Si X
        // Sg0 = -(Si X) Si: a FULL product, not lower-triangle-and-mirror -- like the Ki sandwiches of round 2, the mirrored form of this
        // Sigma^-1 sandwich loses the benign structure of its rounding error: config 3 (jitter 1e-6) had the encoder dense-layer gradient
        // off by 1e-3 against 1e-9 (measured, round 4).  Ssym = c (Sg + Sg^T) is then formed exactly symmetric by the tile-pair kernel:
        // writing 2 c Sg straight from the product keeps an antisymmetric rounding residue that the inducing-point gradients see at
        // 2e-5 (virtual-rank test, m = 256), so the 0.1 ms pass stays.
        GEMM(0, 1, m, m, m, -1.0, six, m, mm, Si, m, mm, 0.0, s.mm1, m, mm, L);            // Sg0 (mm1: X is consumed)
        return SVGP_OK;
    };
    if (do_a) {
        // ubar = ud + gK/2 mu;  mubar = Ki ubar + gK/2 u;  tbar = td + c K mubar: two launches (k_big_gemv_fb)
        hipLaunchKernelGGL(k_big_gemv_fb<1>, dim3((m + 15) / 16, L), dim3(256), (size_t)m * sizeof(real), st, a, (const real*)Ki);
        SVGP_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_big_gemv_fb<2>, dim3((m + 15) / 16, L), dim3(256), (size_t)m * sizeof(real), st, a, (const real*)K);
        SVGP_LAUNCH_CHECK();
        if (x_early) RUNC(x_block());
    }
    if (part == 6) return SVGP_OK;
    if (do_b && !x_early) RUNC(x_block());
    hipLaunchKernelGGL(k_big_fb_ssym, dim3(ntp, ntp, L), dim3(256), 0, st, a);
    SVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_big_sum_channels, dim3(nblk(mm)), dim3(256), 0, st, (int)mm, L, real(1), s.mm1, s.Sgs);
    SVGP_LAUNCH_CHECK();
    }
    if (part == 8) return SVGP_OK;
    // The gradient of Ki is needed for the channel sum only (Ki is shared): Kib = rep_weight (gK/2 sum A + sum ubar mu^T) + Qs + Pbar K,
    // then Ki Kib Ki once.  Pbar, Qs: this rank's row sums from svgp_big_stats (mode 1).
    FinArgs f;
    f.m = m; f.L = L; f.Ltot = c->L; f.geco = SVGP_LOSS_FLAGS(c); f.b_global = c->b_global; f.c = cc; f.N_train = c->N_train;
    f.rep_weight = c->rep_weight; f.state = state;
    f.Asum = s.Asum; f.ubar = s.vec0; f.mu = a.mu; f.Qs = s.Qs; f.PbarK = s.tA; f.Zs = s.Zs; f.mubar = s.vec1; f.t = ws + wl.t + ov;
    f.Sgs = s.Sgs; f.HGs = s.HGs; f.Ki = Ki; f.KiPbar = s.Pbar; f.KiKibKi = s.tA; f.Kib = s.tB; f.Kbar = ws + wl.Kbar;
    f.rank1_late = a.rank1_late; f.vbar = a.vbar;
    if (run_9) {
    GEMM(0, 0, m, m, m, 1.0, K, m, 0, s.Pbar, m, 0, 0.0, s.tA, m, 0, 1);                // K Pbar^T = (Pbar K)^T   (s.Pbar holds Pbar^T)
    hipLaunchKernelGGL(k_big_fb_kib, dim3(nblk(mm)), dim3(256), 0, st, f);             // Kib (tB)
    SVGP_LAUNCH_CHECK();
    GEMM(0, 0, m, m, m, 1.0, Ki, m, 0, s.tB, m, 0, 0.0, s.tC, m, 0, 1);                 // Ki Kib             (tC)
    GEMM(0, 1, m, m, m, 1.0, s.tC, m, 0, Ki, m, 0, 0.0, s.tA, m, 0, 1);                 // Ki Kib Ki          (tA)
    GEMM(0, 1, m, m, m, 1.0, s.Pbar, m, 0, Ki, m, 0, 0.0, s.tB, m, 0, 1);               // Pbar^T Ki = (Ki Pbar)^T   (tB)
    }
    f.KiPbar = s.tB;
    if (run_10) {
    hipLaunchKernelGGL(k_big_fb_final, dim3(nblk(mm)), dim3(256), 0, st, f);
    SVGP_LAUNCH_CHECK();
    }
    return SVGP_OK;
}

int svgp_big_posterior_bwd(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, double* ws, const double* state,
                           void* stream) {
    const int b = c->b, m = c->m, L = c->L;
    const long long mm = (long long)m * m, bm = (long long)b * m;
    const real cc = c->N_train / (double)c->b_global;
    const BigScr s = big_scr(c, wl, ws);
    hipStream_t st = (hipStream_t)stream;
    real* Kn = ws + wl.Kn;
    PbArgs a;
    a.b = b; a.m = m; a.L = L; a.geco = SVGP_LOSS_FLAGS(c); a.c = cc; a.state = state;
    a.y = ws + wl.qnet_mu; a.s2 = ws + wl.qnet_var; a.p_m = ws + wl.p_m; a.p_v = ws + wl.p_v; a.e = ws + wl.e;
    a.d = ws + wl.d; a.g_pv = ws + wl.g_pv; a.g_pm = ws + wl.g_pm; a.mvbar = ws + wl.mvbar;
    a.u = ws + wl.u; a.t = ws + wl.t; a.vbar = ws + wl.vbar; a.R = s.bm; a.kSk = s.bl0; a.kv = s.bl1; a.KnKi = s.KnKi;
    a.KnSi = s.KS; a.sKS = s.sKS; a.qbar = s.qbar;
    a.part = ws + wl.Knbar_part; a.Knbar = ws + wl.Knbar; a.knnbar = ws + wl.knnbar; a.ybar = ws + wl.ybar;
    a.s2bar = ws + wl.s2bar;
    // one (b, m, m, L) product: Kn Si_l comes from the forward pass (svgp_big_posterior_fwd on this workspace), Kn Ki from
    // svgp_big_factor_fwd; the d-term's share arrives through Wbar P^T below
    GEMM(0, 1, b, m, m, 1.0, Kn, m, 0, ws + wl.Ssym, m, mm, 0.0, s.bm, m, bm, L);
    hipLaunchKernelGGL(k_big_rowdot, dim3(nblk((long long)b * L * 64)), dim3(256), 0, st, b, m, L, real(0.5), s.bm, bm, Kn,
                       s.bl0, L, 0);
    SVGP_LAUNCH_CHECK();
    GEMM(0, 1, b, L, m, 1.0, Kn, m, 0, ws + wl.vbar, m, 0, 0.0, s.bl1, L, 0, 1);   // kv = Kn vbar^T
    hipLaunchKernelGGL(k_big_pb_elem, dim3(nblk((long long)b * L)), dim3(256), 0, st, a);
    SVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_big_pb_knbar, dim3(b), dim3(256), 5 * (size_t)L * sizeof(real), st, a);
    SVGP_LAUNCH_CHECK();
    GEMM(0, 0, b, m, m, 1.0, s.Wbar, 2 * m, 0, s.PT, m, 0, 1.0, ws + wl.Knbar, m, 0, 1);   // Knbar += Wbar P^T  (P^T = K Ki; Wbar inside X)
    return SVGP_OK;
}
