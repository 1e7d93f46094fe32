// Titsias branch of mainSVGP.variational_loss (SVGPVAE_model.py:246-259; inside-ELBO assembly :882-883).
//
// Reference, per latent channel l (y = qnet_mu[:, l], var = qnet_var[:, l], j = jitter, b = batch rows):
//     C   = diag(var) + K_nm (K_mm + jI)^-1 K_mn + jI                      (b x b, inverted AND Cholesky-factored)
//     L_2 = -1/2 [ b log 2pi + log det C + y^T C^-1 y + sum_n (k_nn - q_n) / var_n ]
// Here: Woodbury in m x m space with d_n = var_n + j, S2 = sum_n k_n k_n^T / d_n, v2 = sum_n y_n k_n / d_n,
// Sigma2 = K + jI + S2, t2 = Sigma2^-1 v2:
//     log det C  = sum_n log d_n + log det Sigma2 - log det(K + jI)
//     y^T C^-1 y = sum_n y_n^2 / d_n - v2 . t2
// so nothing is b x b, the statistics shard over rows (they ride in the statA all-reduce) and the reverse pass
// needs no exchange of its own (Sb, vb below are functions of reduced quantities).  Hand-derived reverse pass, with
// g2 = d(objective)/d(inside-ELBO):
//     Sb = -g2/2 (Sigma2^-1 + t2 t2^T)           vb = g2 t2
//     dk_n    += sum_l [ 2 Sb_l k_n / d_nl + y_nl vb_l / d_nl ] + g2 (sum_l 1/var_nl) (K + jI)^-1 k_n
//     d(1/d)  =  k_n^T Sb k_n + y_n vb.k_n - g2 y_n^2 / 2
//     dy_n     =  (vb.k_n - g2 y_n) / d_n
//     dvar_n   = -d(1/d) / d_n^2 - g2 / (2 d_n) + g2 (k_nn - q_n) / (2 var_n^2)
//     dk_nn   += -g2/2 sum_l 1/var_nl
//     dK      += sum_l Sb_l + g2/2 [ L (K+jI)^-1 - (K+jI)^-1 (sum_l S_l) (K+jI)^-1 ],   S_l = sum_n k_n k_n^T / var_nl
// (checked against autograd of the literal b x b form to 1e-15 before being written here).
// The stages are compositions of svgp_dgemm_batched / svgp_spd_inverse_batched and element-wise kernels on the
// workspace, valid for every m <= 2048; scratch comes from the large-m path's areas, free between stages.
#include "common.hpp"

namespace {

inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

// P2[n][l] = 1 / (var + j);  PY[n][l] = y / (var + j);  KnP[l][n][i] = Kn[n][i] / (var_nl + j)
__global__ __launch_bounds__(256) void k_tit_weights(int b, int L, real jitter, const real* __restrict__ y,
                                                     const real* __restrict__ s2, real* __restrict__ P2,
                                                     real* __restrict__ PY) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)b * L) return;
    const real pp = real(1) / (s2[e] + jitter);
    P2[e] = pp;
    PY[e] = pp * y[e];
}
__global__ __launch_bounds__(256) void k_tit_scale_rows(int b, int m, int L, const real* __restrict__ P2,
                                                        const real* __restrict__ Kn, real* __restrict__ KnP) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (long long)L * b * m) return;
    const int i = (int)(o % m);
    const long long nl = o / m;
    const int n = (int)(nl % b), l = (int)(nl / b);
    KnP[o] = P2[(size_t)n * L + l] * Kn[(size_t)n * m + i];
}

// Sigma2_l = S2_l + K + j I
__global__ __launch_bounds__(256) void k_tit_sigma(int m, int L, real jitter, const real* __restrict__ S2,
                                                   const real* __restrict__ K, real* __restrict__ Sig) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (long long)L * m * m) return;
    const int ij = (int)(o % ((long long)m * m)), i = ij / m, jj = ij % m;
    Sig[o] = S2[o] + K[ij] + (i == jj ? jitter : real(0));
}

// one workgroup: scal[L + l] = v2_l . t2_l;  scal[2 L] = sum_{n,l} [ log d + y^2 / d + (k_nn - q_n) / var ]
__global__ __launch_bounds__(SVGP_BLOCK) void k_tit_scalars(int b, int m, int L, real jitter, const real* __restrict__ y,
                                                            const real* __restrict__ s2, const real* __restrict__ knn,
                                                            const real* __restrict__ q, const real* __restrict__ v2,
                                                            const real* __restrict__ t2, real* __restrict__ scal) {
    __shared__ real red[SVGP_BLOCK / 64 + 1];
    for (int l = 0; l < L; ++l) {
        real acc = 0;
        for (int i = threadIdx.x; i < m; i += blockDim.x) acc += v2[(size_t)l * m + i] * t2[(size_t)l * m + i];
        const real tot = block_sum(acc, red);
        if (threadIdx.x == 0) scal[L + l] = tot;
        __syncthreads();
    }
    real acc = 0;
    for (long long e = threadIdx.x; e < (long long)b * L; e += blockDim.x) {
        const int n = (int)(e / L);
        const real s = s2[e], d = s + jitter, yy = y[e];
        acc += log(d) + yy * yy / d + recip_no_nan(s) * (knn[n] - q[n]);
    }
    const real tot = block_sum(acc, red);
    if (threadIdx.x == 0) scal[2 * L] = tot;
}

// Sb_l = -g2/2 (Sigma2^-1 + t2 t2^T)
__global__ __launch_bounds__(256) void k_tit_sb(int m, int L, int flags, const real* __restrict__ state,
                                                const real* __restrict__ Si2, const real* __restrict__ t2,
                                                real* __restrict__ Sb) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (long long)L * m * m) return;
    const real g2 = svgp_seed_T(flags, L, state);
    const int l = (int)(o / ((long long)m * m)), ij = (int)(o % ((long long)m * m)), i = ij / m, jj = ij % m;
    Sb[o] = real(-0.5) * g2 * (Si2[o] + t2[(size_t)l * m + i] * t2[(size_t)l * m + jj]);
}

// per (n, l): a = k_n . U_l[n],  c = k_n . t2_l (given);  ybar, s2bar += ...;  coefficient buffers for the row update
__global__ __launch_bounds__(256) void k_tit_rows_nl(int b, int m, int L, int flags, real jitter,
                                                     const real* __restrict__ state, const real* __restrict__ y,
                                                     const real* __restrict__ s2, const real* __restrict__ knn,
                                                     const real* __restrict__ q, const real* __restrict__ Kn,
                                                     const real* __restrict__ U, const real* __restrict__ Cb,
                                                     real* __restrict__ ybar, real* __restrict__ s2bar,
                                                     real* __restrict__ cA, real* __restrict__ cB) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)b * L) return;
    const int n = (int)(e / L), l = (int)(e % L);
    const real g2 = svgp_seed_T(flags, L, state);
    const real* kr = Kn + (size_t)n * m;
    const real* ur = U + ((size_t)l * b + n) * m;
    real a = 0;
    for (int i = 0; i < m; ++i) a += kr[i] * ur[i];
    const real s = s2[e], d = s + jitter, pp = real(1) / d, p = recip_no_nan(s), yy = y[e];
    const real vbk = g2 * Cb[e];
    const real dpp = a + yy * vbk - real(0.5) * g2 * yy * yy;
    ybar[e] += pp * (vbk - g2 * yy);
    s2bar[e] += -pp * pp * dpp - real(0.5) * g2 * pp + real(0.5) * g2 * p * p * (knn[n] - q[n]);
    cA[e] = real(2) * pp;
    cB[e] = g2 * pp * yy;
}

// per (n, i): Knbar += sum_l cA U_l[n][i] + sum_l cB t2_l[i] + 2 qb_n W[n][i];  knnbar[n] -= qb_n,  qb_n = g2/2 sum_l 1/var
__global__ __launch_bounds__(256) void k_tit_rows_ni(int b, int m, int L, int flags, const real* __restrict__ state,
                                                     const real* __restrict__ s2, const real* __restrict__ U,
                                                     const real* __restrict__ t2, const real* __restrict__ W,
                                                     const real* __restrict__ cA, const real* __restrict__ cB,
                                                     real* __restrict__ Knbar, real* __restrict__ knnbar) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (long long)b * m) return;
    const int n = (int)(o / m), i = (int)(o % m);
    const real g2 = svgp_seed_T(flags, L, state);
    real acc = 0, psum = 0;
    for (int l = 0; l < L; ++l) {
        const size_t e = (size_t)n * L + l;
        acc += cA[e] * U[((size_t)l * b + n) * m + i] + cB[e] * t2[(size_t)l * m + i];
        psum += recip_no_nan(s2[e]);
    }
    const real qb = real(0.5) * g2 * psum;
    Knbar[o] += acc + real(2) * qb * W[o];
    if (i == 0) knnbar[n] -= qb;
}

// Ssum = sum_l S_l
__global__ __launch_bounds__(256) void k_tit_sum_l(int mm, int L, const real* __restrict__ S, real* __restrict__ out) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= mm) return;
    real acc = 0;
    for (int l = 0; l < L; ++l) acc += S[(size_t)l * mm + o];
    out[o] = acc;
}
// Kbar += sum_l Sb_l + g2/2 (L Ki - T2)
__global__ __launch_bounds__(256) void k_tit_kbar(int mm, int L, int flags, const real* __restrict__ state,
                                                  const real* __restrict__ Sb, const real* __restrict__ Ki,
                                                  const real* __restrict__ T2, real* __restrict__ Kbar) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= mm) return;
    const real g2 = svgp_seed_T(flags, L, state);
    real acc = 0;
    for (int l = 0; l < L; ++l) acc += Sb[(size_t)l * mm + o];
    Kbar[o] += acc + real(0.5) * g2 * ((real)L * Ki[o] - T2[o]);
}

int layouts(const svgp_mnist_cfg* c, svgp_mnist_ws_layout* wl) {
    int rc = svgp_check_cfg(c);
    if (rc) return rc;
    SVGP_REQUIRE(c->titsias, SVGP_ERR_INVALID, "cfg.titsias is 0: the workspace has no Titsias fields");
    return svgp_mnist_ws_layout_get(c, wl);
}

}  // namespace

#define RUNC(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

extern "C" int svgp_gp_titsias_stats(const svgp_mnist_cfg* c, double* ws, void* stream) {
    svgp_mnist_ws_layout wl;
    RUNC(layouts(c, &wl));
    SVGP_REQUIRE(ws, SVGP_ERR_INVALID, "NULL device pointer");
    const int b = c->b, m = c->m, L = c->L;
    hipStream_t st = (hipStream_t)stream;
    real* P2 = ws + wl.scr_bl;               // (b, L)
    real* PY = P2 + (size_t)b * L;           // (b, L)
    real* KnP = ws + wl.scr_bm;              // (L, b, m)
    hipLaunchKernelGGL(k_tit_weights, dim3(nblk((long long)b * L)), dim3(256), 0, st, b, L, c->jitter, ws + wl.qnet_mu,
                       ws + wl.qnet_var, P2, PY);
    SVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_tit_scale_rows, dim3(nblk((long long)L * b * m)), dim3(256), 0, st, b, m, L, P2, ws + wl.Kn, KnP);
    SVGP_LAUNCH_CHECK();
    // S2_l = Kn^T (Kn / d_l)        v2 = PY^T Kn
    RUNC(svgp_dgemm_batched(1, 0, m, m, b, 1.0, ws + wl.Kn, m, 0, KnP, m, (long long)b * m, 0.0, ws + wl.tit_S2, m,
                            (long long)m * m, L, stream));
    RUNC(svgp_dgemm_batched(1, 0, L, m, b, 1.0, PY, L, 0, ws + wl.Kn, m, 0, 0.0, ws + wl.tit_v2, m, 0, 1, stream));
    return SVGP_OK;
}

extern "C" int svgp_gp_titsias_fwd(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    svgp_mnist_ws_layout wl;
    RUNC(layouts(c, &wl));
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    const int b = c->b, m = c->m, L = c->L;
    hipStream_t st = (hipStream_t)stream;
    real* Si2 = ws + wl.tit_Si;
    real* scal = ws + wl.tit_scal;           // [log det Sigma2 (L) | v2.t2 (L) | row sum (1)]
    hipLaunchKernelGGL(k_tit_sigma, dim3(nblk((long long)L * m * m)), dim3(256), 0, st, m, L, c->jitter, ws + wl.tit_S2,
                       ws + wl.K, Si2);
    SVGP_LAUNCH_CHECK();
    RUNC(svgp_spd_inverse_batched(m, L, Si2, scal, ws + wl.scr_inv, stream));
    // t2_l = Sigma2_l^-1 v2_l
    RUNC(svgp_dgemm_batched(0, 0, m, 1, m, 1.0, Si2, m, (long long)m * m, ws + wl.tit_v2, 1, m, 0.0, ws + wl.tit_t, 1, m, L,
                            stream));
    hipLaunchKernelGGL(k_tit_scalars, dim3(1), dim3(SVGP_BLOCK), 0, st, b, m, L, c->jitter, ws + wl.qnet_mu,
                       ws + wl.qnet_var, ws + wl.knn, ws + wl.q, ws + wl.tit_v2, ws + wl.tit_t, scal);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_gp_titsias_bwd(const svgp_mnist_cfg* c, double* ws, const double* state, void* stream) {
    svgp_mnist_ws_layout wl;
    RUNC(layouts(c, &wl));
    SVGP_REQUIRE(ws && state, SVGP_ERR_INVALID, "NULL device pointer");
    const int b = c->b, m = c->m, L = c->L, flags = SVGP_LOSS_FLAGS(c);
    const long long mm = (long long)m * m;
    hipStream_t st = (hipStream_t)stream;
    real* Sb = ws + wl.scr_mm;                       // (L, m, m)
    real* Ssum = Sb + (size_t)L * mm;                // (m, m)
    real* T1 = Ssum + mm;                            // (m, m)
    real* T2 = T1 + mm;                              // (m, m)        (scr_mm holds 4 L m^2)
    real* U = ws + wl.scr_bm;                        // (L, b, m)
    real* Cb = ws + wl.scr_bl;                       // (b, L)
    real* W = ws + wl.Knbar_part;                    // (b, m)   Knbar_part is dead after svgp_gp_posterior_bwd
    real* cA = W + (size_t)b * m;                    // (b, L)   (Knbar_part holds L b m >= b m + 2 b L for m >= 2 ... )
    real* cB = cA + (size_t)b * L;
    SVGP_REQUIRE((long long)L * b * m >= (long long)b * m + 2LL * b * L, SVGP_ERR_UNSUPPORTED,
                 "Titsias reverse pass needs (L - 1) m >= 2 L scratch elements per row (L=%d m=%d)", L, m);
    hipLaunchKernelGGL(k_tit_sb, dim3(nblk((long long)L * mm)), dim3(256), 0, st, m, L, flags, state, ws + wl.tit_Si,
                       ws + wl.tit_t, Sb);
    SVGP_LAUNCH_CHECK();
    // U_l = Kn Sb_l;  Cb = Kn t2^T;  W = Kn Ki
    RUNC(svgp_dgemm_batched(0, 0, b, m, m, 1.0, ws + wl.Kn, m, 0, Sb, m, mm, 0.0, U, m, (long long)b * m, L, stream));
    RUNC(svgp_dgemm_batched(0, 1, b, L, m, 1.0, ws + wl.Kn, m, 0, ws + wl.tit_t, m, 0, 0.0, Cb, L, 0, 1, stream));
    RUNC(svgp_dgemm_batched(0, 0, b, m, m, 1.0, ws + wl.Kn, m, 0, ws + wl.Ki, m, 0, 0.0, W, m, 0, 1, stream));
    hipLaunchKernelGGL(k_tit_rows_nl, dim3(nblk((long long)b * L)), dim3(256), 0, st, b, m, L, flags, c->jitter, state,
                       ws + wl.qnet_mu, ws + wl.qnet_var, ws + wl.knn, ws + wl.q, ws + wl.Kn, U, Cb, ws + wl.ybar,
                       ws + wl.s2bar, cA, cB);
    SVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_tit_rows_ni, dim3(nblk((long long)b * m)), dim3(256), 0, st, b, m, L, flags, state,
                       ws + wl.qnet_var, U, ws + wl.tit_t, W, cA, cB, ws + wl.Knbar, ws + wl.knnbar);
    SVGP_LAUNCH_CHECK();
    // Kbar += sum_l Sb_l + g2/2 (L Ki - Ki (sum_l S_l) Ki)
    hipLaunchKernelGGL(k_tit_sum_l, dim3(nblk(mm)), dim3(256), 0, st, (int)mm, L * svgp_stat_parts(c), ws + wl.S, Ssum);   // sum over channels and row partials
    SVGP_LAUNCH_CHECK();
    RUNC(svgp_dgemm_batched(0, 0, m, m, m, 1.0, ws + wl.Ki, m, 0, Ssum, m, 0, 0.0, T1, m, 0, 1, stream));
    RUNC(svgp_dgemm_batched(0, 0, m, m, m, 1.0, T1, m, 0, ws + wl.Ki, m, 0, 0.0, T2, m, 0, 1, stream));
    hipLaunchKernelGGL(k_tit_kbar, dim3(nblk(mm)), dim3(256), 0, st, (int)mm, L, flags, state, Sb, ws + wl.Ki, T2,
                       ws + wl.Kbar);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
