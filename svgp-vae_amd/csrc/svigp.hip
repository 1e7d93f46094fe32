// Deep SVIGP_Hensman baseline on rotated MNIST (SVIGP_Hensman_model.py:14-339).
// Free variational parameters per latent channel (loc mu_l (m), scale A_l (m,m), S_l = A_l A_l^T), a Gaussian
// likelihood noise, mnistSVGP's kernel and the mnistVAE decoder.  The GP block in O(b m^2 + L m^3):
//   Ki = (K + jI)^-1, W = K_nm Ki, mean_vectors Z = W M^T                                   (:156-160)
//   sum_l L_3 = -p/2 [ L sum_n (k_nn - q_n) + <sum_l S_l, W^T W> ],  p = 1/noise              (:176-196)
//   sum_l KL  = 1/2 [ L logdet(K+jI) - sum_l logdet(S_l+jI) - L m + <Ki, sum_l S_l> + sum_l mu_l.Ki mu_l ]   (:164-173)
//   elbo = -b K_pix log noise - b K_pix log(2 pi)/2 - recon_sq / (2 noise^2) + sum L_3 - (b/N) sum KL   (:255-286)
// and the hand-derived reverse pass of -elbo.  Every contraction is a batched float64 MFMA GEMM (linalg.hip), the
// inverses are blocked Gauss-Jordan; the kernels here are the element-wise / reduction glue.
#include "common.hpp"

extern "C" int svgp_dgemm_batched(int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda,
                                  long long strideA, const double* B, int ldb, long long strideB, double beta,
                                  double* C, int ldc, long long strideC, int batch, void* stream);
extern "C" size_t svgp_spd_inverse_workspace_elems(int m, int batch);
extern "C" int svgp_spd_inverse_batched(int m, int batch, double* A, double* logdet, double* work, void* stream);

namespace {

inline int nb(long long n) { return (int)((n + 255) / 256); }
inline size_t al8(size_t n) { return (n + 7) & ~(size_t)7; }

struct Lay {
    size_t Ki, ldK, W, U, S, Sji, ldS, H, Ssum, scal, T1, T2, Wbar, Tb, Kibar, Sbar, inv, total;
};
Lay layout(int b, int m, int L) {
    Lay o;
    size_t off = 0;
    auto take = [&](size_t n) { size_t r = off; off += al8(n); return r; };
    const size_t mm = (size_t)m * m;
    o.Ki = take(mm); o.ldK = take(1); o.W = take((size_t)b * m); o.U = take((size_t)L * m); o.S = take(L * mm);
    o.Sji = take(L * mm); o.ldS = take(L); o.H = take(mm); o.Ssum = take(mm); o.scal = take(8); o.T1 = take(mm);
    o.T2 = take(mm); o.Wbar = take((size_t)b * m); o.Tb = take((size_t)b * m); o.Kibar = take(mm); o.Sbar = take(L * mm);
    const size_t i1 = svgp_spd_inverse_workspace_elems(m, 1), iL = svgp_spd_inverse_workspace_elems(m, L);
    o.inv = take(i1 > iL ? i1 : iL);
    o.total = off;
    return o;
}

// out[l] = in[l] + jitter I
__global__ void k_add_jitter(int m, long long tot, real jitter, const real* __restrict__ in, real* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tot) return;
    const long long o = i % ((long long)m * m);
    out[i] = in[i] + ((o / m) == (o % m) ? jitter : real(0));
}

// one workgroup: Ssum = sum_l S_l; the scalar terms of the forward pass
__global__ __launch_bounds__(256) void k_svigp_scalars(int b, int m, int L, int n_pix, const real* __restrict__ Kn,
                                                       const real* __restrict__ knn, const real* __restrict__ W,
                                                       const real* __restrict__ Ki, const real* __restrict__ S,
                                                       const real* __restrict__ H, const real* __restrict__ loc,
                                                       const real* __restrict__ U, const real* __restrict__ ldK,
                                                       const real* __restrict__ ldS, const real* __restrict__ noise,
                                                       real* __restrict__ Ssum, real* __restrict__ scal) {
    __shared__ real red[16];
    const int mm = m * m;
    real kq = 0, trSH = 0, trKiS = 0, muu = 0, lds = 0;
    for (long long i = threadIdx.x; i < (long long)b * m; i += blockDim.x) kq -= W[i] * Kn[i];
    for (int n = threadIdx.x; n < b; n += blockDim.x) kq += knn[n];
    for (int o = threadIdx.x; o < mm; o += blockDim.x) {
        real s = 0;
        for (int l = 0; l < L; ++l) s += S[(size_t)l * mm + o];
        Ssum[o] = s;
        trSH += s * H[o];
        trKiS += s * Ki[o];
    }
    for (int i = threadIdx.x; i < L * m; i += blockDim.x) muu += loc[i] * U[i];
    for (int l = threadIdx.x; l < L; l += blockDim.x) lds += ldS[l];
    kq = block_sum(kq, red); trSH = block_sum(trSH, red); trKiS = block_sum(trKiS, red); muu = block_sum(muu, red);
    lds = block_sum(lds, red);
    if (threadIdx.x == 0) {
        const real nz = *noise, p = real(1) / nz;
        scal[0] = real(-0.5) * p * ((real)L * kq + trSH);                                        // sum_l L_3
        scal[1] = real(0.5) * ((real)L * *ldK - lds - (real)L * (real)m + trKiS + muu);          // sum_l KL
        scal[2] = real(0.5) * p * p * (real)n_pix;   // d(-elbo)/d recon = f * [2 (recon - x) / n_pix]
        scal[3] = kq; scal[4] = trSH;
    }
}

__global__ void k_scale_dev(long long n, const real* __restrict__ f, real* __restrict__ x) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= *f;
}
// Wbar = ZbM + p (W Ssum) - (p L / 2) Kn        (T holds W Ssum on entry)
__global__ void k_svigp_wbar(long long n, int L, const real* __restrict__ noise, const real* __restrict__ T,
                             const real* __restrict__ Kn, real* __restrict__ Wbar) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const real p = real(1) / *noise;
    Wbar[i] += p * T[i] - real(0.5) * p * (real)L * Kn[i];
}
// Knbar += -(p L / 2) W ; knnbar = p L / 2
__global__ void k_svigp_knbar(int b, int m, int L, const real* __restrict__ noise, const real* __restrict__ W,
                              real* __restrict__ Knbar, real* __restrict__ knnbar) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const real h = real(0.5) * (real)L / *noise;
    if (i < (long long)b * m) Knbar[i] -= h * W[i];
    if (i < b) knnbar[i] = h;
}
// Kibar += c2/2 (Ssum + M^T M);  d_mu = Mbar + c2 U;  Sbar_l = p/2 H + c2/2 (Ki - Sji_l)
__global__ void k_svigp_mid(int m, int L, real c2, const real* __restrict__ noise, const real* __restrict__ Ssum,
                            const real* __restrict__ MtM, const real* __restrict__ U, const real* __restrict__ H,
                            const real* __restrict__ Ki, const real* __restrict__ Sji, real* __restrict__ Kibar,
                            real* __restrict__ d_mu, real* __restrict__ Sbar) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long mm = (long long)m * m;
    const real p = real(1) / *noise;
    if (i < mm) Kibar[i] += real(0.5) * c2 * (Ssum[i] + MtM[i]);
    if (i < (long long)L * m) d_mu[i] += c2 * U[i];
    if (i < (long long)L * mm) {
        const long long o = i % mm;
        Sbar[i] = real(0.5) * p * H[o] + real(0.5) * c2 * (Ki[o] - Sji[i]);
    }
}
// Kbar += (c2 L / 2) Ki ; d_noise
__global__ void k_svigp_fin(int m, int L, real c2, long long bK, int n_part, const real* __restrict__ part_sums,
                            const real* __restrict__ noise, const real* __restrict__ Ki, const real* __restrict__ scal,
                            real* __restrict__ Kbar, real* __restrict__ d_noise) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long long)m * m) Kbar[i] += real(0.5) * c2 * (real)L * Ki[i];
    if (i == 0) {
        real rs = 0;
        for (int g = 0; g < n_part; ++g) rs += part_sums[4 * g + 2];
        const real nz = *noise;
        // -elbo = bK log nz + rs / (2 nz^2) - sum L_3 + ..;  -sum L_3 = p/2 (L kq + trSH), dp/dnz = -1/nz^2
        *d_noise = (real)bK / nz - rs / (nz * nz * nz) - real(0.5) * ((real)L * scal[3] + scal[4]) / (nz * nz);
    }
}
// out (7): [elbo, recon_loss (per pixel count), KL_term, inside_elbo, 0, inside_recon, inside_kl]
__global__ void k_svigp_assemble(int b, int n_pix, real c2, int n_part, const real* __restrict__ part_sums,
                                 const real* __restrict__ noise, const real* __restrict__ scal, real* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    real rs = 0;
    for (int g = 0; g < n_part; ++g) rs += part_sums[4 * g + 2];
    const real nz = *noise, bK = (real)b * (real)n_pix, inside = scal[0] - c2 * scal[1];
    out[0] = -bK * log(nz) - real(0.5) * bK * real(SVGP_LOG_2PI) - real(0.5) * rs / (nz * nz) + inside;
    out[1] = rs / (real)n_pix; out[2] = inside; out[3] = inside; out[4] = 0; out[5] = scal[0]; out[6] = scal[1];
}

}  // namespace

#define RUNC(x)              \
    do {                     \
        int rc_ = (x);       \
        if (rc_) return rc_; \
    } while (0)
#define GEMM(...) RUNC(svgp_dgemm_batched(__VA_ARGS__, stream))

static int check(int b, int b_global, int m, int L) {
    SVGP_REQUIRE(b >= 1 && b_global >= b && m >= 1 && L >= 1, SVGP_ERR_INVALID, "bad shape b=%d b_global=%d m=%d L=%d", b,
                 b_global, m, L);
    SVGP_REQUIRE(m <= SVGP_M_LIMIT && L <= 64, SVGP_ERR_UNSUPPORTED, "m=%d L=%d outside m <= %d, L <= 64", m, L, SVGP_M_LIMIT);
    return SVGP_OK;
}

extern "C" long long svgp_svigp_workspace_elems(int b, int m, int L) {
    if (b < 1 || m < 1 || L < 1) return -1;
    return (long long)layout(b, m, L).total;
}

// K (m,m), Kn (b,m), knn (b): kernel matrices; loc (L,m), scale (L,m,m), noise (1): the variational parameters and the
// likelihood noise.  Z (b,L) = mean vectors (the decoder's input).  The scalar terms stay in the workspace.
extern "C" int svgp_svigp_fwd(int b, int b_global, int m, int L, int n_pix, double jitter, const double* K,
                              const double* Kn, const double* knn, const double* loc, const double* scale,
                              const double* noise, double* Z, double* ws, void* stream) {
    RUNC(check(b, b_global, m, L));
    SVGP_REQUIRE(K && Kn && knn && loc && scale && noise && Z && ws, SVGP_ERR_INVALID, "NULL device pointer");
    const Lay o = layout(b, m, L);
    hipStream_t st = (hipStream_t)stream;
    const long long mm = (long long)m * m;
    real *Ki = ws + o.Ki, *W = ws + o.W, *U = ws + o.U, *S = ws + o.S, *Sji = ws + o.Sji, *H = ws + o.H;
    hipLaunchKernelGGL(k_add_jitter, dim3(nb(mm)), dim3(256), 0, st, m, mm, jitter, K, Ki);
    SVGP_LAUNCH_CHECK();
    RUNC(svgp_spd_inverse_batched(m, 1, Ki, ws + o.ldK, ws + o.inv, stream));
    GEMM(0, 0, b, m, m, 1.0, Kn, m, 0, Ki, m, 0, 0.0, W, m, 0, 1);                       // W = Kn Ki
    GEMM(0, 0, L, m, m, 1.0, loc, m, 0, Ki, m, 0, 0.0, U, m, 0, 1);                      // U_l = Ki mu_l
    GEMM(0, 1, b, L, m, 1.0, W, m, 0, loc, m, 0, 0.0, Z, L, 0, 1);                       // Z = W M^T
    GEMM(0, 1, m, m, m, 1.0, scale, m, mm, scale, m, mm, 0.0, S, m, mm, L);              // S_l = A_l A_l^T
    hipLaunchKernelGGL(k_add_jitter, dim3(nb(mm * L)), dim3(256), 0, st, m, mm * L, jitter, S, Sji);
    SVGP_LAUNCH_CHECK();
    RUNC(svgp_spd_inverse_batched(m, L, Sji, ws + o.ldS, ws + o.inv, stream));
    GEMM(1, 0, m, m, b, 1.0, W, m, 0, W, m, 0, 0.0, H, m, 0, 1);                         // H = W^T W
    hipLaunchKernelGGL(k_svigp_scalars, dim3(1), dim3(256), 0, st, b, m, L, n_pix, Kn, knn, W, Ki, S, H, loc, U,
                       ws + o.ldK, ws + o.ldS, noise, ws + o.Ssum, ws + o.scal);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// Zbar (b,L): d[(recon - x)^2 sum / n_pix]/dZ as the mnistVAE decoder's reverse pass produces it in beta-ELBO mode (scaled
// here in place by f = n_pix / (2 noise^2)); part_sums: the decoder's (n_part,4) partial sums whose column 2 holds the
// squared reconstruction error.  Outputs: Kbar (m,m), Knbar (b,m), knnbar (b), d_loc (L,m), d_scale (L,m,m), d_noise (1).
extern "C" int svgp_svigp_bwd(int b, int b_global, int m, int L, int n_pix, double N_train, const double* Kn,
                              const double* loc, const double* scale, const double* noise, double* Zbar,
                              const double* part_sums, int n_part, double* Kbar, double* Knbar, double* knnbar,
                              double* d_loc, double* d_scale, double* d_noise, double* ws, void* stream) {
    RUNC(check(b, b_global, m, L));
    SVGP_REQUIRE(Kn && loc && scale && noise && Zbar && part_sums && Kbar && Knbar && knnbar && d_loc && d_scale &&
                     d_noise && ws, SVGP_ERR_INVALID, "NULL device pointer");
    SVGP_REQUIRE(N_train > 0 && n_part >= 1, SVGP_ERR_INVALID, "bad N_train / n_part");
    const Lay o = layout(b, m, L);
    hipStream_t st = (hipStream_t)stream;
    const long long mm = (long long)m * m, bm = (long long)b * m;
    const real c2 = (real)b_global / N_train;
    real *Ki = ws + o.Ki, *W = ws + o.W, *U = ws + o.U, *Sji = ws + o.Sji, *H = ws + o.H, *Ssum = ws + o.Ssum;
    real *T1 = ws + o.T1, *T2 = ws + o.T2, *Wbar = ws + o.Wbar, *Tb = ws + o.Tb, *Kibar = ws + o.Kibar, *Sbar = ws + o.Sbar;
    hipLaunchKernelGGL(k_scale_dev, dim3(nb((long long)b * L)), dim3(256), 0, st, (long long)b * L, ws + o.scal + 2, Zbar);
    SVGP_LAUNCH_CHECK();
    GEMM(1, 0, L, m, b, 1.0, Zbar, L, 0, W, m, 0, 0.0, d_loc, m, 0, 1);                  // Mbar = Zb^T W
    GEMM(0, 0, b, m, L, 1.0, Zbar, L, 0, loc, m, 0, 0.0, Wbar, m, 0, 1);                 // Wbar = Zb M
    GEMM(0, 0, b, m, m, 1.0, W, m, 0, Ssum, m, 0, 0.0, Tb, m, 0, 1);                     // W Ssum
    hipLaunchKernelGGL(k_svigp_wbar, dim3(nb(bm)), dim3(256), 0, st, bm, L, noise, Tb, Kn, Wbar);
    SVGP_LAUNCH_CHECK();
    GEMM(0, 0, b, m, m, 1.0, Wbar, m, 0, Ki, m, 0, 0.0, Knbar, m, 0, 1);                 // Knbar = Wbar Ki
    hipLaunchKernelGGL(k_svigp_knbar, dim3(nb(bm > b ? bm : b)), dim3(256), 0, st, b, m, L, noise, W, Knbar, knnbar);
    SVGP_LAUNCH_CHECK();
    GEMM(1, 0, m, m, b, 1.0, Kn, m, 0, Wbar, m, 0, 0.0, Kibar, m, 0, 1);                 // Kibar = Kn^T Wbar
    GEMM(1, 0, m, m, L, 1.0, loc, m, 0, loc, m, 0, 0.0, T1, m, 0, 1);                    // M^T M
    const long long nmid = mm * L > (long long)L * m ? mm * L : (long long)L * m;
    hipLaunchKernelGGL(k_svigp_mid, dim3(nb(nmid)), dim3(256), 0, st, m, L, c2, noise, Ssum, T1, U, H, Ki, Sji, Kibar,
                       d_loc, Sbar);
    SVGP_LAUNCH_CHECK();
    GEMM(0, 0, m, m, m, 1.0, Ki, m, 0, Kibar, m, 0, 0.0, T2, m, 0, 1);                   // Ki Kibar
    GEMM(0, 0, m, m, m, -1.0, T2, m, 0, Ki, m, 0, 0.0, Kbar, m, 0, 1);                   // Kbar = -Ki Kibar Ki
    GEMM(0, 0, m, m, m, 2.0, Sbar, m, mm, scale, m, mm, 0.0, d_scale, m, mm, L);         // d_A_l = 2 Sbar_l A_l
    hipLaunchKernelGGL(k_svigp_fin, dim3(nb(mm)), dim3(256), 0, st, m, L, c2, (long long)b * n_pix, n_part, part_sums,
                       noise, Ki, ws + o.scal, Kbar, d_noise);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// out (7) = [elbo, recon_loss / n_pix, KL_term, inside_elbo, 0, inside_recon, inside_kl] (SVIGP_Hensman_model.py:262-289)
extern "C" int svgp_svigp_assemble(int b, int b_global, int m, int L, int n_pix, double N_train, const double* noise,
                                   const double* part_sums, int n_part, const double* ws, double* out, void* stream) {
    RUNC(check(b, b_global, m, L));
    SVGP_REQUIRE(noise && part_sums && ws && out && n_part >= 1 && N_train > 0, SVGP_ERR_INVALID, "bad argument");
    const Lay o = layout(b, m, L);
    hipLaunchKernelGGL(k_svigp_assemble, dim3(1), dim3(64), 0, (hipStream_t)stream, b, n_pix, (real)b_global / N_train,
                       n_part, part_sums, noise, ws + o.scal, out);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// x[0..n) *= *f  (f on the device): the decoder's weight gradients take the same factor f as Zbar
extern "C" int svgp_scale_by_device_scalar(long long n, const double* f, double* x, void* stream) {
    SVGP_REQUIRE(n >= 1 && f && x, SVGP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_scale_dev, dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, n, f, x);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" long long svgp_svigp_scale_offset(int b, int m, int L) { return (long long)layout(b, m, L).scal + 2; }
