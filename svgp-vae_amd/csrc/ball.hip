// Moving-ball experiment pieces (BALL_experiment.py): everything that is not the
// shared sparse-GP stages.  The GP block of build_SVGPVAE_elbo_graph (SVGPVAE_model.py:638-715) runs on the
// channel-batched stages of gp_kernels.hip with rows = the tmax frames and channels = the videos of the batch
// (every video has the same time stamps 1..tmax, :663-664, so K_mm and K_nm are shared), cfg.kl_form = 1,
// cfg.clip_pv = 2, N_train = tmax; one workspace per latent coordinate (svgp_x, svgp_y).
//   SE kernel on scalar times + VJP                       SVGP.__init__ :60, kernel.matrix calls :80-86
//   MLP bias / tanh layers + reverse                      VAE_utils.py:9-96 (the matmuls are svgp_dgemm_batched)
//   encoder head exp / clip, (batch,tmax,4) <-> (tmax,batch) channel layout   VAE_utils.py:50-55, SVGPVAE_model.py:670-671
//   Bernoulli reconstruction term + d/d logits            SVGPVAE_model.py:697-700
//   per-video ELBO assembly + scalar epilogue             SVGPVAE_model.py:677-705, BALL_experiment.py:116-123
//   exact per-video GP of the Pearce baseline + VJP       GPVAE_Pearce_model.py:8-86
#include "common.hpp"

namespace {

inline int nb256(long long n) { return (int)((n + 255) / 256); }

// ---------------------------------------------------------------------------------------------------------
// SE kernel matrices on scalar inputs: K (m,m), Kn (T,m), knn (T) = 1
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_se1d_fwd(int T, int m, const real* __restrict__ x, const real* __restrict__ z,
                                                  const real* __restrict__ ls, real* __restrict__ K,
                                                  real* __restrict__ Kn, real* __restrict__ knn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const real il2 = real(-0.5) / (*ls * *ls);
    if (i < T * m) {
        const real d = x[i / m] - z[i % m];
        Kn[i] = exp(d * d * il2);
    } else if (i < T * m + m * m) {
        const int o = i - T * m;
        const real d = z[o / m] - z[o % m];
        K[o] = exp(d * d * il2);
    } else if (i < T * m + m * m + T) {
        knn[i - T * m - m * m] = real(1);
    }
}

// one workgroup: d_z[i] = sum_j (Kbar_ij + Kbar_ji) K_ij (z_j - z_i)/l^2 + sum_n Knbar_ni Kn_ni (x_n - z_i)/l^2
//                d_l    = sum_ij Kbar_ij K_ij (z_i - z_j)^2 / l^3 + sum_ni Knbar_ni Kn_ni (x_n - z_i)^2 / l^3
__global__ __launch_bounds__(256) void k_se1d_bwd(int T, int m, const real* __restrict__ x, const real* __restrict__ z,
                                                  const real* __restrict__ ls, const real* __restrict__ Kbar,
                                                  const real* __restrict__ Knbar, real* __restrict__ d_z,
                                                  real* __restrict__ d_ls) {
    __shared__ real red[16];
    const real l = *ls, il2 = real(1) / (l * l), il3 = il2 / l;
    real dl = 0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const real zi = z[i];
        real dz = 0;
        for (int j = 0; j < m; ++j) {
            const real d = zi - z[j], k = exp(real(-0.5) * d * d * il2);
            dz -= (Kbar[i * m + j] + Kbar[j * m + i]) * k * d * il2;
            dl += Kbar[i * m + j] * k * d * d * il3;
        }
        for (int n = 0; n < T; ++n) {
            const real d = x[n] - zi, g = Knbar[(size_t)n * m + i] * exp(real(-0.5) * d * d * il2);
            dz += g * d * il2;
            dl += g * d * d * il3;
        }
        d_z[i] = dz;
    }
    dl = block_sum(dl, red);
    if (threadIdx.x == 0) *d_ls = dl;
}

// ---------------------------------------------------------------------------------------------------------
// dense-layer glue: x = act(x + bias) in place; dpre = dout * act'(out) in place + column sums
// ---------------------------------------------------------------------------------------------------------
__global__ void k_bias_act(long long tot, int C, int act, const real* __restrict__ bias, real* __restrict__ x) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tot) return;
    const real v = x[i] + bias[i % C];
    x[i] = act == 1 ? tanh(v) : v;
}
// grid (ceil(C/256), NCH): thread = column, rows strided by NCH; part (NCH, C)
#define ACT_NCH 32
__global__ __launch_bounds__(256) void k_act_bwd_colsum(int rows, int C, int act, const real* __restrict__ out,
                                                        real* __restrict__ dout, real* __restrict__ part) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    real s = 0;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        const size_t o = (size_t)r * C + c;
        real dv = dout[o];
        if (act == 1) { const real ov = out[o]; dv *= (real(1) - ov * ov); dout[o] = dv; }
        s += dv;
    }
    part[(size_t)blockIdx.y * C + c] = s;
}
__global__ void k_colsum_final(int C, int nch, const real* __restrict__ part, real* __restrict__ db) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    real s = 0;
    for (int g = 0; g < nch; ++g) s += part[(size_t)g * C + c];
    db[c] = s;
}

// ---------------------------------------------------------------------------------------------------------
// encoder head: h (B*T, 4) + bias -> per coordinate c in {x, y}: mu, var_raw = exp(.), var = clip(var_raw) in the
// (T, B) channel layout of the GP workspaces
// ---------------------------------------------------------------------------------------------------------
struct HeadPtrs { real* mu[2]; real* var_raw[2]; real* var[2]; };
__global__ void k_ball_head_fwd(int B, int T, int clip, const real* __restrict__ bias, const real* __restrict__ h,
                                HeadPtrs o) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // i = (b*T + t)*2 + c
    if (i >= B * T * 2) return;
    const int c = i & 1, r = i >> 1, b = r / T, t = r % T;
    const size_t e = (size_t)t * B + b;
    o.mu[c][e] = h[(size_t)r * 4 + c] + bias[c];
    const real vr = exp(h[(size_t)r * 4 + 2 + c] + bias[2 + c]);
    o.var_raw[c][e] = vr;
    o.var[c][e] = clip ? fmin(fmax(vr, 1e-6), 1e3) : vr;
}
struct HeadBwdPtrs { const real* var_raw[2]; const real* ybar[2]; const real* s2bar[2]; };
__global__ void k_ball_head_bwd(int B, int T, int clip, HeadBwdPtrs p, real* __restrict__ dh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T * 2) return;
    const int c = i & 1, r = i >> 1, b = r / T, t = r % T;
    const size_t e = (size_t)t * B + b;
    const real vr = p.var_raw[c][e];
    const bool pass = !clip || (vr >= 1e-6 && vr <= 1e3);     // tf.clip_by_value gradient mask
    dh[(size_t)r * 4 + c] = p.ybar[c][e];
    dh[(size_t)r * 4 + 2 + c] = pass ? p.s2bar[c][e] * vr : real(0);
}
// latent samples (T,B) x 2 -> (B*T, 2) and the reverse for their gradient
__global__ void k_ball_pack_z(int B, int T, const real* __restrict__ zx, const real* __restrict__ zy,
                              real* __restrict__ z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T * 2) return;
    const int c = i & 1, r = i >> 1, b = r / T, t = r % T;
    z[i] = (c ? zy : zx)[(size_t)t * B + b];
}
__global__ void k_ball_unpack_zbar(int B, int T, const real* __restrict__ dz, real* __restrict__ zbx,
                                   real* __restrict__ zby) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T * 2) return;
    const int c = i & 1, r = i >> 1, b = r / T, t = r % T;
    (c ? zby : zbx)[(size_t)t * B + b] = dz[i];
}

// ---------------------------------------------------------------------------------------------------------
// Bernoulli reconstruction: per frame  row_recon = -sum_pix xent(label, logit),  pred = sigmoid(logit),
// dlogits = scale (sigmoid - label)   (scale = 1/batch: loss = -mean_b elbo_b).  One workgroup per frame.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sigmoid_xent(int P, real scale, const real* __restrict__ logits,
                                                      const real* __restrict__ labels, real* __restrict__ pred,
                                                      real* __restrict__ row_recon, real* __restrict__ dlogits) {
    __shared__ real red[16];
    const size_t base = (size_t)blockIdx.x * P;
    real s = 0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const real x = logits[base + i], zl = labels[base + i];
        const real ex = exp(-fabs(x));
        s += fmax(x, real(0)) - x * zl + log1p(ex);          // tf.nn.sigmoid_cross_entropy_with_logits
        const real sg = x >= 0 ? real(1) / (real(1) + ex) : ex / (real(1) + ex);
        if (pred) pred[base + i] = sg;
        if (dlogits) dlogits[base + i] = scale * (sg - zl);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) row_recon[blockIdx.x] = -s;
}

// ---------------------------------------------------------------------------------------------------------
// per-video ELBO pieces.  One workgroup per video, thread = frame.  out (7, B):
//   [elbo, recon, KL_term, inside_elbo, ce_term, inside_recon, inside_kl]      (SVGPVAE_model.py:677-705)
// ---------------------------------------------------------------------------------------------------------
struct BallChan {
    const real* y; const real* s2; const real* p_m; const real* p_v; const real* d; const real* KL;
    const real* knn; const real* q; const real* tit_scal; const real* ldK;
};
struct BallAsm {
    int B, T, titsias;
    real jitter;
    BallChan ch[2];
    const real* row_recon; const real* state;
    real* out;
};
__global__ __launch_bounds__(64) void k_ball_assemble(BallAsm a) {
    __shared__ real red[4];
    const int b = blockIdx.x, B = a.B, T = a.T;
    real l3 = 0, ce = 0, rows = 0, rec = 0;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        rec += a.row_recon[(size_t)b * T + t];
        for (int c = 0; c < 2; ++c) {
            const BallChan& h = a.ch[c];
            const size_t e = (size_t)t * B + b;
            const real s2 = h.s2[e], p = recip_no_nan(s2), y = h.y[e], ls2 = log(s2), dm = h.p_m[e] - y;
            l3 += real(-0.5) * (p * h.d[e] + ls2 + real(SVGP_LOG_2PI));
            ce += real(0.5) * (real(SVGP_LOG_2PI) + ls2 + (h.p_v[e] + dm * dm) * p);     // -gauss_cross_entropy
            if (a.titsias) {
                const real dj = s2 + a.jitter;
                rows += log(dj) + y * y / dj + p * (h.knn[t] - h.q[t]) + real(SVGP_LOG_2PI);
            }
        }
    }
    l3 = block_sum(l3, red); ce = block_sum(ce, red); rows = block_sum(rows, red); rec = block_sum(rec, red);
    if (threadIdx.x != 0) return;
    real in_rec, in_kl = 0;
    if (a.titsias) {
        in_rec = real(-0.5) * rows;
        for (int c = 0; c < 2; ++c)
            in_rec += real(-0.5) * (a.ch[c].tit_scal[b] - *a.ch[c].ldK - a.ch[c].tit_scal[B + b]);
    } else {
        in_rec = l3;
        for (int c = 0; c < 2; ++c) {
            // reference: every video carries the batch-wide scalar 1/2 sum_l tr(Ki A_l A_l) (SVGPVAE_model.py:135-137);
            // the stage kernels store KL_l with L tr(Ki A_l A_l) (same sum over videos) and the traces behind it
            const real* KL = a.ch[c].KL;
            real tot = 0;
            for (int l = 0; l < B; ++l) tot += KL[B + l];
            in_kl += KL[b] - real(0.5) * (real)B * KL[B + b] + real(0.5) * tot;
        }
    }
    const real inside = in_rec - in_kl, klt = ce + inside, beta = a.state[SVGP_ST_BETA];
    real* o = a.out;
    o[b] = rec + beta * klt; o[B + b] = rec; o[2 * B + b] = klt; o[3 * B + b] = inside; o[4 * B + b] = ce;
    o[5 * B + b] = in_rec; o[6 * B + b] = in_kl;
}
// means over videos -> state scalars; advances the Adam step counter and the RNG counter
__global__ __launch_bounds__(64) void k_ball_finalize(int B, int did_adam, long long rng_advance,
                                                      const real* __restrict__ out, real* __restrict__ st) {
    __shared__ real red[4];
    const int slot[7] = {SVGP_ST_ELBO, SVGP_ST_RECON_LOSS, SVGP_ST_KL_TERM, SVGP_ST_INSIDE_ELBO, SVGP_ST_CE_TERM,
                         SVGP_ST_INSIDE_RECON, SVGP_ST_INSIDE_KL};
    for (int k = 0; k < 7; ++k) {
        real s = 0;
        for (int b = threadIdx.x; b < B; b += blockDim.x) s += out[(size_t)k * B + b];
        s = block_sum(s, red);
        if (threadIdx.x == 0) st[slot[k]] = s / (real)B;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (did_adam) st[SVGP_ST_ADAM_T] += real(1);
        st[SVGP_ST_RNG_CTR] += (real)rng_advance;
    }
}

__global__ void k_state_add(int slot, real v, real* __restrict__ st) { st[slot] += v; }

// binary ball frames from pixel-space centres (utils.py:177-187): frame[i][j] = (i - x)^2 + (j - y)^2 < r^2
__global__ void k_ball_rasterize(long long tot, int px, int py, real rr, const real* __restrict__ paths,
                                 real* __restrict__ vid) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tot) return;
    const int j = (int)(i % py), ii = (int)((i / py) % px);
    const long long f = i / ((long long)px * py);
    const real dx = (real)ii - paths[2 * f], dy = (real)j - paths[2 * f + 1];
    vid[i] = (dx * dx + dy * dy < rr) ? real(1) : real(0);
}

}  // namespace

#define REQ_PTRS(...)                                                                             \
    do {                                                                                          \
        const void* ps_[] = {__VA_ARGS__};                                                        \
        for (const void* q_ : ps_) SVGP_REQUIRE(q_ != nullptr, SVGP_ERR_INVALID, "NULL device pointer"); \
    } while (0)

extern "C" int svgp_se1d_kernel_matrix_fwd(int T, int m, const double* x, const double* z, const double* ls, double* K,
                                           double* Kn, double* knn, void* stream) {
    SVGP_REQUIRE(T >= 1 && m >= 1, SVGP_ERR_INVALID, "need T >= 1, m >= 1 (T=%d m=%d)", T, m);
    REQ_PTRS(x, z, ls, K, Kn, knn);
    hipLaunchKernelGGL(k_se1d_fwd, dim3(nb256((long long)T * m + (long long)m * m + T)), dim3(256), 0,
                       (hipStream_t)stream, T, m, x, z, ls, K, Kn, knn);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_se1d_kernel_matrix_bwd(int T, int m, const double* x, const double* z, const double* ls,
                                           const double* Kbar, const double* Knbar, double* d_z, double* d_ls,
                                           void* stream) {
    SVGP_REQUIRE(T >= 1 && m >= 1, SVGP_ERR_INVALID, "need T >= 1, m >= 1 (T=%d m=%d)", T, m);
    REQ_PTRS(x, z, ls, Kbar, Knbar, d_z, d_ls);
    hipLaunchKernelGGL(k_se1d_bwd, dim3(1), dim3(256), 0, (hipStream_t)stream, T, m, x, z, ls, Kbar, Knbar, d_z, d_ls);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_bias_act_fwd(long long rows, int C, int act, const double* bias, double* x, void* stream) {
    SVGP_REQUIRE(rows >= 1 && C >= 1 && (act == 0 || act == 1), SVGP_ERR_INVALID, "bad argument");
    REQ_PTRS(bias, x);
    hipLaunchKernelGGL(k_bias_act, dim3(nb256(rows * C)), dim3(256), 0, (hipStream_t)stream, rows * C, C, act, bias, x);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_act_bwd_bias(int rows, int C, int act, const double* out, double* dout, double* part, double* db,
                                 void* stream) {
    SVGP_REQUIRE(rows >= 1 && C >= 1 && (act == 0 || act == 1), SVGP_ERR_INVALID, "bad argument");
    REQ_PTRS(dout, part, db);
    SVGP_REQUIRE(act == 0 || out != nullptr, SVGP_ERR_INVALID, "activation output is NULL");
    hipLaunchKernelGGL(k_act_bwd_colsum, dim3((C + 255) / 256, ACT_NCH), dim3(256), 0, (hipStream_t)stream, rows, C, act,
                       out, dout, part);
    SVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_colsum_final, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, ACT_NCH, part, db);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_act_bwd_bias_scratch_elems(int C) { return ACT_NCH * C; }

extern "C" int svgp_ball_head_fwd(int B, int T, int clip, const double* bias, const double* h, double* mu_x,
                                  double* var_raw_x, double* var_x, double* mu_y, double* var_raw_y, double* var_y,
                                  void* stream) {
    SVGP_REQUIRE(B >= 1 && T >= 1, SVGP_ERR_INVALID, "bad shape");
    REQ_PTRS(bias, h, mu_x, var_raw_x, var_x, mu_y, var_raw_y, var_y);
    HeadPtrs o;
    o.mu[0] = mu_x; o.mu[1] = mu_y; o.var_raw[0] = var_raw_x; o.var_raw[1] = var_raw_y; o.var[0] = var_x; o.var[1] = var_y;
    hipLaunchKernelGGL(k_ball_head_fwd, dim3(nb256((long long)B * T * 2)), dim3(256), 0, (hipStream_t)stream, B, T, clip,
                       bias, h, o);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_ball_head_bwd(int B, int T, int clip, const double* var_raw_x, const double* ybar_x,
                                  const double* s2bar_x, const double* var_raw_y, const double* ybar_y,
                                  const double* s2bar_y, double* dh, void* stream) {
    SVGP_REQUIRE(B >= 1 && T >= 1, SVGP_ERR_INVALID, "bad shape");
    REQ_PTRS(var_raw_x, ybar_x, s2bar_x, var_raw_y, ybar_y, s2bar_y, dh);
    HeadBwdPtrs p;
    p.var_raw[0] = var_raw_x; p.var_raw[1] = var_raw_y; p.ybar[0] = ybar_x; p.ybar[1] = ybar_y;
    p.s2bar[0] = s2bar_x; p.s2bar[1] = s2bar_y;
    hipLaunchKernelGGL(k_ball_head_bwd, dim3(nb256((long long)B * T * 2)), dim3(256), 0, (hipStream_t)stream, B, T, clip,
                       p, dh);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_ball_pack_z(int B, int T, const double* zx, const double* zy, double* z, void* stream) {
    SVGP_REQUIRE(B >= 1 && T >= 1, SVGP_ERR_INVALID, "bad shape");
    REQ_PTRS(zx, zy, z);
    hipLaunchKernelGGL(k_ball_pack_z, dim3(nb256((long long)B * T * 2)), dim3(256), 0, (hipStream_t)stream, B, T, zx, zy, z);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_ball_unpack_zbar(int B, int T, const double* dz, double* zbar_x, double* zbar_y, void* stream) {
    SVGP_REQUIRE(B >= 1 && T >= 1, SVGP_ERR_INVALID, "bad shape");
    REQ_PTRS(dz, zbar_x, zbar_y);
    hipLaunchKernelGGL(k_ball_unpack_zbar, dim3(nb256((long long)B * T * 2)), dim3(256), 0, (hipStream_t)stream, B, T, dz,
                       zbar_x, zbar_y);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_sigmoid_xent(int rows, int P, double scale, const double* logits, const double* labels, double* pred,
                                 double* row_recon, double* dlogits, void* stream) {
    SVGP_REQUIRE(rows >= 1 && P >= 1, SVGP_ERR_INVALID, "bad shape");
    REQ_PTRS(logits, labels, row_recon);
    hipLaunchKernelGGL(k_sigmoid_xent, dim3(rows), dim3(256), 0, (hipStream_t)stream, P, scale, logits, labels, pred,
                       row_recon, dlogits);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// cfg_x / cfg_y: the two GP workspaces' configurations (b = tmax rows, L = batch videos); out: (7, batch)
extern "C" int svgp_ball_elbo_assemble(const svgp_mnist_cfg* cx, const double* ws_x, const double* ws_y,
                                       const double* row_recon, const double* state, double* out, void* stream) {
    svgp_mnist_ws_layout wl;
    int rc = svgp_mnist_ws_layout_get(cx, &wl);
    if (rc) return rc;
    REQ_PTRS(ws_x, ws_y, row_recon, state, out);
    BallAsm a;
    a.B = cx->L; a.T = cx->b; a.titsias = cx->titsias; a.jitter = cx->jitter;
    const double* w[2] = {ws_x, ws_y};
    for (int c = 0; c < 2; ++c) {
        BallChan& h = a.ch[c];
        h.y = w[c] + wl.qnet_mu; h.s2 = w[c] + wl.qnet_var; h.p_m = w[c] + wl.p_m; h.p_v = w[c] + wl.p_v;
        h.d = w[c] + wl.d; h.KL = w[c] + wl.KL; h.knn = w[c] + wl.knn; h.q = w[c] + wl.q;
        h.tit_scal = w[c] + wl.tit_scal; h.ldK = w[c] + wl.ldK;
    }
    a.row_recon = row_recon; a.state = state; a.out = out;
    hipLaunchKernelGGL(k_ball_assemble, dim3(a.B), dim3(64), 0, (hipStream_t)stream, a);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

extern "C" int svgp_ball_finalize(int B, int did_adam, long long rng_advance, const double* out, double* state,
                                  void* stream) {
    SVGP_REQUIRE(B >= 1, SVGP_ERR_INVALID, "bad shape");
    REQ_PTRS(out, state);
    hipLaunchKernelGGL(k_ball_finalize, dim3(1), dim3(64), 0, (hipStream_t)stream, B, did_adam, rng_advance, out, state);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// state[slot] += v on the stream (e.g. a fresh Philox counter for the second latent coordinate's samples)
extern "C" int svgp_state_add(double* state, int slot, double v, void* stream) {
    SVGP_REQUIRE(state && slot >= 0 && slot < SVGP_STATE_LEN, SVGP_ERR_INVALID, "bad state slot %d", slot);
    hipLaunchKernelGGL(k_state_add, dim3(1), dim3(1), 0, (hipStream_t)stream, slot, v, state);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// paths (frames, 2) pixel-space centres -> vid (frames, px, py) binary frames of a ball of radius r
extern "C" int svgp_ball_rasterize(long long frames, int px, int py, double r, const double* paths, double* vid,
                                   void* stream) {
    SVGP_REQUIRE(frames >= 1 && px >= 1 && py >= 1, SVGP_ERR_INVALID, "bad shape");
    REQ_PTRS(paths, vid);
    const long long tot = frames * px * py;
    hipLaunchKernelGGL(k_ball_rasterize, dim3(nb256(tot)), dim3(256), 0, (hipStream_t)stream, tot, px, py, r * r, paths, vid);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
