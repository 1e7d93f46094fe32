// Shared declarations for libsvgpvae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/svgpvae_hip.h"

typedef double real;

#define SVGP_BLOCK 256
#define SVGP_MAX_PART 256      // max workgroups that write weight-gradient partials (= CUs)
#define SVGP_M_MAX 64          // up to here the m x m stages stay LDS-resident (gp_kernels.hip)
#define SVGP_M_LIMIT 2048      // beyond SVGP_M_MAX: global-memory matrices + batched MFMA GEMMs (gp_large.hip)
#define SVGP_CHOL_INVERSE_MIN_M 512   // spd inverse: fused 32-block Gauss-Jordan sweep below, potrf + potri from here on
#define SVGP_LOG_2PI 1.8378770664093453

void svgp_set_error(const char* fmt, ...);

#define SVGP_CHECK_HIP(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            svgp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                     \
            return SVGP_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

#define SVGP_REQUIRE(cond, code, ...)   \
    do {                                \
        if (!(cond)) {                  \
            svgp_set_error(__VA_ARGS__); \
            return (code);              \
        }                               \
    } while (0)

// api.hip: fork / join of the library-owned side branch of a caller stream (one per (device, stream))
int svgp_side_branch_fork(void* main_stream, void** side_stream_out, int k = 1);
// cholesky.hip: svgp_potrf_batched that zeroes only the 128 columns right of the diagonal (library-internal consumers)
int svgp_potri_batched_wide(int m, int batch, double* A, const double* potrf_work, double* work, void* stream);
int svgp_potrf_batched_band(int m, int batch, double* A, int lda, long long strideA, double* logdet, double* work, void* stream);
int svgp_side_branch_join(void* main_stream, int k = 1);
// linalg.hip: svgp_dgemm_splitk with a second factor for the result rows >= row2
int svgp_dgemm_splitk_rows2(int ta, int tb, int M, int N, int K, double alpha, double alpha2, int row2, const double* A, int lda,
                            const double* B, int ldb, double beta, double* C, int ldc, double* scratch, long long scratch_elems,
                            void* stream);
// gp_kernels.hip: parts 5 / 6 / 7 of the large-m forward factor stage (see gp_large.hip svgp_big_factor_fwd)
int svgp_gp_factor_fwd_part(const svgp_mnist_cfg* c, double* ws, void* stream, int part);
int svgp_mnist_step_phase_deferred(const svgp_mnist_cfg* c, int phase, double* theta, const double* images,
                                   const double* aux, const double* eps, double* ws, double* state, double* adam_m,
                                   double* adam_v, void* stream);

// Loss seeds of the reverse passes.  `flags` = cfg.geco | cfg.titsias << 1 (SVGP_LOSS_FLAGS).
//   seed_T : d(minimised objective)/d(KL_term): GECO -1 (SVGPVAE_model.py:913), beta-ELBO -beta/L (:925); also
//            the seed of the cross-entropy term and, for Titsias, of sum_l L_2
//   seed_3 : seed of the Hensman L_3 terms  (0 for Titsias: L_2 is handled by gp_titsias.hip)
//   seed_K : seed of the Hensman KL terms   (0 for Titsias)
#define SVGP_LOSS_FLAGS(c) ((c)->geco | ((c)->titsias ? 2 : 0))
#ifdef __HIPCC__
__device__ __forceinline__ real svgp_seed_T(int flags, int L, const real* state) {
    return (flags & 1) ? real(-1) : -state[SVGP_ST_BETA] / (real)L;
}
__device__ __forceinline__ real svgp_seed_3(int flags, real gT) { return (flags & 2) ? real(0) : gT; }
// d loss / d p_v: cross-entropy part + sampling part z = p_m + eps sqrt(clip(p_v)).  clip_pv 1: p_v itself is clipped
// to [1e-4, 100] (SVGPVAE_model.py:891-892), the mask kills both parts; 2: only the sample clips, to [1e-4, 1000] (:693)
__device__ __forceinline__ real svgp_gpv(int clip_pv, real gT, real p, real zbar, real eps, real pv) {
    const real ce = real(0.5) * gT * p, sm = zbar * eps / (real(2) * sqrt(pv));
    if (clip_pv == 1) return (pv > 1e-4 && pv < 100.0) ? ce + sm : real(0);
    if (clip_pv == 2) return (pv > 1e-4 && pv < 1000.0) ? ce + sm : ce;
    return ce + sm;
}
__device__ __forceinline__ real svgp_seed_K(int flags, real gT, real b_over_N) {
    return (flags & 2) ? real(0) : -gT * b_over_N;
}
#endif

#define SVGP_LAUNCH_CHECK() SVGP_CHECK_HIP(hipGetLastError())

#ifdef __HIPCC__
// MFMA traits of the two arithmetic types (16 x 16 x 4 forms; A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15])
typedef double svgp_d4_t __attribute__((ext_vector_type(4)));
typedef float svgp_f4_t __attribute__((ext_vector_type(4)));
template <typename TC> struct SvgpMfma;
template <> struct SvgpMfma<double> {
    typedef svgp_d4_t acc_t;
    typedef double2 pair_t;
    __device__ static __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int q, int e) { return q + 4 * e; }       // C/D: row = (lane >> 4) + 4 reg
};
template <> struct SvgpMfma<float> {
    typedef svgp_f4_t acc_t;
    typedef float2 pair_t;
    __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int q, int e) { return 4 * q + e; }       // C/D: row = 4 (lane >> 4) + reg
};
#endif

// Row partials of the statistics launches (LDS-resident path): the rows of every channel are split over this many
// workgroups, each writing its own partial block; consumers add the partials on load.  A function of the row CAPACITY
// (like every offset of the layout).  One block when the batch is sharded over ranks: the statistics blocks are then
// all-reduced, and 4x the bytes on the xGMI ring (~+14 us per exchange at 8 GPUs) costs more than the split saves.
#define SVGP_STAT_PARTS 4
inline int svgp_stat_parts(const svgp_mnist_cfg* c) {
    const int cap = c->b_cap > 0 ? c->b_cap : c->b;
    return (c->m <= SVGP_M_MAX && cap >= 128 && !c->single_stat_block) ? SVGP_STAT_PARTS : 1;
}

int svgp_check_cfg(const svgp_mnist_cfg* c);

static inline int svgp_n_part(const svgp_mnist_cfg* c) {
    return c->b < SVGP_MAX_PART ? c->b : SVGP_MAX_PART;
}
static inline int svgp_n_rowblk(const svgp_mnist_cfg* c) { return (c->b + 63) / 64; }
// per-sample kernels: thread (row, i) with SVGP_BLOCK / m rows per workgroup
static inline int svgp_rows_per_block(const svgp_mnist_cfg* c) { return c->m >= SVGP_BLOCK ? 1 : SVGP_BLOCK / c->m; }
static inline int svgp_n_postblk(const svgp_mnist_cfg* c) {
    const int rb = svgp_rows_per_block(c);
    return (c->b + rb - 1) / rb;
}

// number of (L3, CE) partial pairs the per-sample forward writes for this configuration
static inline int svgp_n_post_actual(const svgp_mnist_cfg* c) {
    return c->m > SVGP_M_MAX ? (c->b * c->L + SVGP_BLOCK - 1) / SVGP_BLOCK : c->L * svgp_n_postblk(c);
}

// linalg.hip: fused one-launch-per-block-step inverse of nmain + nextra matrices (m < SVGP_CHOL_INVERSE_MIN_M)
int svgp_spd_inverse_fused(int m, int nmain, double* A, double* logdet, int nextra, double* Ae, double* logdet_e,
                           double* work, void* stream);

// Extended GEMM epilogue (float64 storage), applied per stored element of the output tile:
//   C  = alpha acc + beta C + g1 E + d1 I          C2 = a2 acc + g2 E + d2 I   (optional second output, leading dimension ldc2)
// E: extra matrix (leading dimension lde, batch stride se: 0 = shared by the batch); with the mirrored store of a symmetric
// product E is read at the mirrored position.  Replaces the element-wise passes over (L, m, m) arrays of the large-m GP block.
struct svgp_gemm_epi {
    const double* E = nullptr; int lde = 0; long long se = 0;
    double g1 = 0, d1 = 0;
    double* C2 = nullptr; long long sc2 = 0; int ldc2 = 0;      // ldc2 = 0: the leading dimension of C
    double a2 = 0, g2 = 0, d2 = 0;
    // alpha is multiplied by *alpha_dev (a DEVICE scalar: a loss seed that lives in the device state vector) when set
    const double* alpha_dev = nullptr;
    // E is exactly symmetric (a mirrored-store product): the mirrored store of a symmetric output reuses the value read at (i, j)
    // instead of reading E at (j, i) with a stride of lde
    int e_sym = 0;
};
int svgp_dgemm_tri_batched(int tri, int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda,
                           long long strideA, const double* B, int ldb, long long strideB, double beta, double* C, int ldc,
                           long long strideC, int batch, void* stream, const svgp_gemm_epi* epi = nullptr);

// wk (optional): weights of the contraction index, op(B)[k][:] *= wk[k * ldw + l * strideW] while staged (C symmetric iff A = B)
int svgp_dgemm_symout_batched(int f32c, int ta, int tb, int M, int K, double alpha, const double* A, int lda, long long strideA,
                              const double* B, int ldb, long long strideB, double beta, double* C, int ldc, long long strideC,
                              int batch, void* stream, const double* wk = nullptr, int ldw = 0, long long strideW = 0,
                              const svgp_gemm_epi* epi = nullptr);
int svgp_dgemm_bsub_batched(int f32c, int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda, long long strideA,
                            const double* B, int ldb, long long strideB, const double* Bsub, double beta, double* C, int ldc,
                            long long strideC, int batch, void* stream);
int svgp_dgemm_epi_batched(int f32c, int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda, long long strideA,
                           const double* B, int ldb, long long strideB, double beta, double* C, int ldc, long long strideC,
                           int batch, void* stream, const svgp_gemm_epi* epi);

// large-m implementations (gp_large.hip)
int svgp_big_stats(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, double* ws, const double* state, int mode,
                   void* stream);
int svgp_big_factor_fwd(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, double* ws, void* stream, int l0, int nl,
                        int part = 0);
int svgp_big_posterior_fwd(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, const double* eps, double* ws,
                           double* state, void* stream);
int svgp_big_factor_bwd(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, double* ws, const double* state,
                        void* stream, int l0, int nl, int part = 0);
int svgp_big_posterior_bwd(const svgp_mnist_cfg* c, const svgp_mnist_ws_layout& wl, double* ws, const double* state,
                           void* stream);

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ real wave_sum(real x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;  // valid in lane 0
}

// Workgroup barrier that orders LDS traffic ONLY.  `__syncthreads()` is a workgroup-scope fence as well: every wave first waits for
// ALL its outstanding global loads and stores (s_waitcnt vmcnt(0)), so a `store activations; barrier; next layer` sequence exposes a
// full store round trip (~0.7 us on MI355X) on a chain whose next step reads LDS only.  Use this where no wave of the workgroup
// reads, after the barrier, global memory another wave wrote before it (the compiler still waits for a global LOAD at its first use).
__device__ __forceinline__ void svgp_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Sum over the workgroup (blockDim.x multiple of 64, <= 1024); result valid in thread 0.
// `red` is an LDS scratch of >= 16 reals.  Fixed order -> deterministic.
__device__ __forceinline__ real block_sum(real x, real* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    x = wave_sum(x);
    __syncthreads();
    if (lane == 0) red[w] = x;
    __syncthreads();
    real s = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

// block_sum for a workgroup of which only the first nt threads (a multiple of 64) are alive: rider workgroups with fewer threads
// than the launch they ride in (the other waves have exited; s_barrier counts the surviving waves only)
__device__ __forceinline__ real block_sum_nt(real x, real* red, int nt) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = nt >> 6;
    x = wave_sum(x);
    __syncthreads();
    if (lane == 0) red[w] = x;
    __syncthreads();
    real s = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

__device__ __forceinline__ real elu_f(real x) { return x > 0 ? x : expm1(x); }
// derivative of ELU expressed through its OUTPUT: out > 0 -> 1, else exp(pre) = out + 1
__device__ __forceinline__ real elu_grad_from_out(real out) { return out > 0 ? real(1) : out + real(1); }

// Philox4x32-10 -> one N(0,1) double (Box-Muller on two 53-bit uniforms... 2 x 32-bit + 2 x 32-bit)
__device__ __forceinline__ real svgp_philox_normal(unsigned long long ctr, unsigned long long idx) {
    unsigned int c0 = (unsigned int)idx, c1 = (unsigned int)(idx >> 32), c2 = (unsigned int)ctr,
                 c3 = (unsigned int)(ctr >> 32);
    unsigned int k0 = 0x5356u, k1 = 0x47505641u;   // fixed key
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned int)p1;
        const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned int)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const real u1 = ((real)(((unsigned long long)c0 << 21) ^ (unsigned long long)(c1 >> 11)) + real(0.5)) *
                    (real(1) / real(9007199254740992.0));   // (0,1), 53 bits
    const real u2 = ((real)(((unsigned long long)c2 << 21) ^ (unsigned long long)(c3 >> 11)) + real(0.5)) *
                    (real(1) / real(9007199254740992.0));
    return sqrt(real(-2) * log(u1)) * cos(real(6.283185307179586) * u2);
}

__device__ __forceinline__ real recip_no_nan(real x) { return x == real(0) ? real(0) : real(1) / x; }

// mnistSVGP.kernel_matrix arguments (SVGPVAE_model.py:427-476), shared by gp_kernels.hip and the encoder launch that
// carries the kernel-matrix build in its spare workgroups
struct SvgpKernArgs {
    int b, m, M, n_obj, normalize;
    const real* aux;   // (b, 2+M)
    const real* ip;    // (m, 2+M)
    const real* ov;    // (n_obj, M) or unused
    const real* ls;    // scalar
    const real* amp;   // scalar
};
inline SvgpKernArgs svgp_make_kern_args(const svgp_mnist_cfg* c, const svgp_mnist_param_layout& pl, const double* theta,
                                        const double* aux) {
    SvgpKernArgs a;
    a.b = c->b; a.m = c->m; a.M = c->M; a.n_obj = c->n_obj; a.normalize = c->normalize_obj;
    a.aux = aux; a.ip = theta + pl.ip; a.ov = theta + pl.ov; a.ls = theta + pl.l_GP; a.amp = theta + pl.amplitude;
    return a;
}

#ifdef __HIPCC__
__device__ __forceinline__ const real* svgp_obj_row(const SvgpKernArgs& a, int n) {
    return a.n_obj > 0 ? a.ov + (size_t)((long long)a.aux[(size_t)n * (2 + a.M)]) * a.M
                       : a.aux + (size_t)n * (2 + a.M) + 2;
}
__device__ __forceinline__ real svgp_view_k(real d, real a2, real inv_l2) {
    const real s = sin(real(0.5) * d);
    return a2 * exp(real(-2) * s * s * inv_l2);
}
__device__ __forceinline__ real svgp_dotM(const real* x, const real* y, int M) {
    real s = 0;
    for (int k = 0; k < M; ++k) s += x[k] * y[k];
    return s;
}
// element idx of [K_nm (b m) | K_mm (m m) | k_nn (b)]; idx beyond: nothing
__device__ __forceinline__ void svgp_km_fwd_element(const SvgpKernArgs& a, long long idx, real* __restrict__ K,
                                                    real* __restrict__ Kn, real* __restrict__ knn) {
    const long long nbm = (long long)a.b * a.m, nmm = (long long)a.m * a.m;
    const int st = 2 + a.M;
    const real amp = *a.amp, ls = *a.ls, a2 = amp * amp, inv_l2 = real(1) / (ls * ls);
    if (idx < nbm) {
        const int n = (int)(idx / a.m), j = (int)(idx % a.m);
        const real* on = svgp_obj_row(a, n);
        const real* oj = a.ip + (size_t)j * st + 2;
        real D = svgp_dotM(on, oj, a.M);
        if (a.normalize) D /= sqrt(svgp_dotM(on, on, a.M)) * sqrt(svgp_dotM(oj, oj, a.M));
        Kn[idx] = svgp_view_k(a.aux[(size_t)n * st + 1] - a.ip[(size_t)j * st + 1], a2, inv_l2) * D;
    } else if (idx < nbm + nmm) {
        const long long o = idx - nbm;
        const int i = (int)(o / a.m), j = (int)(o % a.m);
        const real* oi = a.ip + (size_t)i * st + 2;
        const real* oj = a.ip + (size_t)j * st + 2;
        real D = svgp_dotM(oi, oj, a.M);
        if (a.normalize) D /= sqrt(svgp_dotM(oi, oi, a.M)) * sqrt(svgp_dotM(oj, oj, a.M));
        K[o] = svgp_view_k(a.ip[(size_t)i * st + 1] - a.ip[(size_t)j * st + 1], a2, inv_l2) * D;
    } else if (idx < nbm + nmm + a.b) {
        const int n = (int)(idx - nbm - nmm);
        const real* on = svgp_obj_row(a, n);
        knn[n] = a.normalize ? a2 : a2 * svgp_dotM(on, on, a.M);
    }
}

// dynamic LDS of the scatter workgroups: the 256 staged d_on rows while M <= 32 (64 KB); wider rows are read from global memory
// (staged only when the whole batch is ONE chunk of 256 rows: with several chunks every workgroup would stage all b rows, 64 KB a
// chunk at M = 32, to use the ~b / n_obj rows per table row that match -- config 3: 40 us for the scatter workgroups, and their
// dynamic LDS held every workgroup of the closing reduction they ride in to two per CU; reading the matching rows directly: 5 us)
static inline bool svgp_km_scatter_staged(int b, int M) { return M <= 32 && b <= 256; }
static inline size_t svgp_km_scatter_lds(int b, int M) { return svgp_km_scatter_staged(b, M) ? (size_t)256 * M * sizeof(real) : 0; }
// Object-table scatter of the kernel-matrix VJP, one workgroup of SVGP_BLOCK threads (dynamic LDS: 256 * M doubles at
// `dbuf`, M <= 32).  Workgroups [0, nblk - 1): element o = blk * 256 + tid of the (n_obj, M) table gradient = sum of the d_on
// rows whose id matches, in row order (duplicate ids sum deterministically): per chunk of 256 staged rows a bit mask per
// table row is built with LDS atomicOr and walked with ffs.  Workgroup nblk - 1: the amplitude / length-scale partial sums.
__device__ __forceinline__ void svgp_km_scatter_block(int blk, int nblk, int b, int M, int n_obj,
                                                      const real* __restrict__ aux, int n_gp_part, int train_gp,
                                                      int train_ov, const real* __restrict__ d_on,
                                                      const real* __restrict__ part_gp, real* __restrict__ d_ov,
                                                      real* __restrict__ d_ls, real* __restrict__ d_amp) {
    extern __shared__ __align__(16) real svgp_dyn_lds[];
    __shared__ real sred[16];
    if (blk == nblk - 1) {
        real sa = 0, sl = 0;
        for (int i = threadIdx.x; i < n_gp_part; i += blockDim.x) { sa += part_gp[i * 2]; sl += part_gp[i * 2 + 1]; }
        sa = block_sum(sa, sred);
        sl = block_sum(sl, sred);
        if (threadIdx.x == 0) { *d_amp = train_gp ? sa : real(0); *d_ls = train_gp ? sl : real(0); }
        return;
    }
    real* dbuf = svgp_dyn_lds;      // 256 x M
    __shared__ unsigned mask[256][8];    // per table row of this workgroup: which of the 256 staged batch rows match
    const int o = blk * blockDim.x + threadIdx.x;
    const bool act = o < n_obj * M;
    const int r_first = (blk * (int)blockDim.x) / M, r_last = (blk * (int)blockDim.x + (int)blockDim.x - 1) / M;
    const int r = act ? o / M : -1, k = act ? o % M : 0, st = 2 + M;
    real acc = 0;
    for (int n0 = 0; n0 < b; n0 += 256) {
        const int cnt = min(256, b - n0);
        __syncthreads();
        for (int t = threadIdx.x; t < 256 * 8; t += blockDim.x) (&mask[0][0])[t] = 0u;
        const bool staged = M <= 32 && b <= 256;
        if (staged)
            for (int t = threadIdx.x; t < cnt * M; t += blockDim.x) dbuf[t] = d_on[(size_t)n0 * M + t];
        __syncthreads();
        // set bits are order-independent (atomicOr), the sums below walk them in increasing row order: the result
        // is the row-order sum whatever the execution order
        if ((int)threadIdx.x < cnt) {
            const int id = (int)aux[(size_t)(n0 + threadIdx.x) * st];
            if (id >= r_first && id <= r_last) atomicOr(&mask[id - r_first][threadIdx.x >> 5], 1u << (threadIdx.x & 31));
        }
        __syncthreads();
        if (train_ov && act) {
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                unsigned bits = mask[r - r_first][w];
                while (bits) {
                    const int n = w * 32 + __ffs(bits) - 1;
                    acc += staged ? dbuf[n * M + k] : d_on[(size_t)(n0 + n) * M + k];
                    bits &= bits - 1;
                }
            }
        }
    }
    if (act) d_ov[o] = acc;
}
#endif
