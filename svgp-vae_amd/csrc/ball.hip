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
// one workgroup per 16-column tile over ALL rows: thread = (column c16, row lane rl of 16); the 16 row lanes are added
// through LDS in fixed order, so the column sums need no partial buffer and no second launch
__global__ __launch_bounds__(256) void k_act_bwd_colsum(int rows, int C, int act, const real* __restrict__ out,
                                                        real* __restrict__ dout, real* __restrict__ db) {
    __shared__ real sh[16][17];
    const int c16 = threadIdx.x & 15, rl = threadIdx.x >> 4, c = blockIdx.x * 16 + c16;
    real s = 0;
    if (c < C) {
#pragma unroll 4
        for (int r = rl; r < rows; r += 16) {
            const size_t o = (size_t)r * C + c;
            real dv = dout[o];
            if (act == 1) { const real ov = out[o]; dv *= (real(1) - ov * ov); dout[o] = dv; }
            s += dv;
        }
    }
    sh[rl][c16] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        real t = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][c16];
        db[c] = t;
    }
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


// ---------------------------------------------------------------------------------------------------------
// Exact per-video GP regression of the Pearce baseline (build_1d_gp, GPVAE_Pearce_model.py:8-86), X_test = X.
// One workgroup per (video, latent coordinate); T <= 64 frames, all T x T matrices in LDS.
//   A = K_SE(l) + diag(var);  alpha = A^-1 y;  lhood = -1/2 (n log 2pi + y.alpha + log det A)
//   p_m = K alpha;  p_v = 1 - diag(K A^-1 K);  z = p_m + eps sqrt(p_v)        (:201-202 of the ELBO builder)
// idx != NULL (neural-process context sets, :121-155): the n points are times[idx[b][0..n)] and only lhood is produced.
// Values of y / var / p_m / p_v / eps / z / zbar are in the (T, B) channel layout of the head kernels.
// ---------------------------------------------------------------------------------------------------------
struct PearceArgs {
    int B, T, n, use_rng;
    const real* times; const int* idx;          // idx (B, n) or NULL
    const real* ls[2];
    const real* y[2]; const real* s2[2];
    const real* eps_in[2]; const real* state;
    real* Ai;                                    // (2, B, n, n)
    real* alpha;                                 // (2, B, n)
    real* p_m[2]; real* p_v[2]; real* eps[2]; real* z[2];
    real* lh;                                    // (2, B)
    real* ce;                                    // (2, B)   sum_t -gauss_cross_entropy
    real* row_ce;                                // (2, T, B) the summands (neural-process target sums)
    const real* tmask;                           // (B, T) 1 = target frame, or NULL (reverse: weights of the CE terms)
    // reverse
    const real* zbar[2];
    real* ybar[2]; real* s2bar[2];
    real* dl_part;                               // (2, B) length-scale gradient partials
    real seed_lh_scale;                          // lhood enters the loss with seed gT * seed_lh_scale (NP context: -1)
    int accumulate;                              // reverse: add into ybar / s2bar (context sets) instead of writing
};

// Thread layout of the T x T work below (256 threads), CW = 32 (n <= 32) or 64 columns: column j = tid & (CW-1), row
// lane i0 = tid / CW of RL = 256 / CW; a thread owns the elements (i0 + RL r, j).  No integer division in the loops.
#define PEARCE_IJ for (int i = i0; i < n; i += RL) if (j < n)

// in-place Gauss-Jordan inverse of an SPD n x n LDS matrix (leading dimension ld, n <= 64), no pivoting; returns log det
template <int CW>
__device__ real lds_spd_inverse(real* A, int ld, int n, real* pivbuf) {
    constexpr int RL = 256 / CW, RMAX = CW / RL;
    const int j = threadIdx.x & (CW - 1), i0 = threadIdx.x / CW;
    real logdet = 0;
    for (int k = 0; k < n; ++k) {
        __syncthreads();
        const real piv = A[k * ld + k], ip = real(1) / piv;
        const real rowk = j < n ? A[k * ld + j] : real(0);
        real f[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) { const int i = i0 + RL * r; f[r] = i < n ? A[i * ld + k] * ip : real(0); }
        if (threadIdx.x == 0) pivbuf[k] = piv;           // the logs are taken once, in parallel, after the sweep
        __syncthreads();
        if (j < n) {
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                const int i = i0 + RL * r;
                if (i < n && i != k) A[i * ld + j] = (j == k) ? -f[r] : A[i * ld + j] - f[r] * rowk;
            }
            if (i0 == 0) A[k * ld + j] = (j == k) ? ip : rowk * ip;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {                              // n <= 64: one wave, fixed-order shuffle sum
        const real lg = wave_sum((int)threadIdx.x < n ? log(pivbuf[threadIdx.x]) : real(0));
        if (threadIdx.x == 0) pivbuf[0] = lg;
    }
    __syncthreads();
    logdet = pivbuf[0];
    __syncthreads();
    return logdet;
}

__device__ __forceinline__ real pearce_time(const PearceArgs& a, int b, int i) {
    return a.idx ? a.times[a.idx[(size_t)b * a.n + i]] : a.times[i];
}
__device__ __forceinline__ size_t pearce_elem(const PearceArgs& a, int b, int i) {
    const int t = a.idx ? a.idx[(size_t)b * a.n + i] : i;
    return (size_t)t * a.B + b;
}

template <int CW>
__global__ __launch_bounds__(256) void k_pearce_fwd(PearceArgs a) {
    extern __shared__ __align__(16) real smem[];
    constexpr int RL = 256 / CW;
    const int b = blockIdx.x, c = blockIdx.y, n = a.n, ld = n + 1;
    const int j = threadIdx.x & (CW - 1), i0 = threadIdx.x / CW;
    real* K = smem;               // n x ld
    real* A = K + n * ld;         // -> Ai
    real* W = A + n * ld;         // Ai K
    real* yv = W + n * ld;        // n
    real* al = yv + n;            // n
    real* tv = al + n;            // n
    real* red = tv + n;           // 16
    const real l = *a.ls[c], il2 = real(-0.5) / (l * l);
    for (int i = threadIdx.x; i < n; i += blockDim.x) { tv[i] = pearce_time(a, b, i); yv[i] = a.y[c][pearce_elem(a, b, i)]; }
    __syncthreads();
    PEARCE_IJ {
        const real d = tv[i] - tv[j], k = exp(d * d * il2);
        K[i * ld + j] = k;
        A[i * ld + j] = k + (i == j ? a.s2[c][pearce_elem(a, b, i)] : real(0));
    }
    const real logdet = lds_spd_inverse<CW>(A, ld, n, al);       // al (n reals) is free until alpha is formed
    const size_t om = ((size_t)c * a.B + b) * n * n, ov = ((size_t)c * a.B + b) * n;
    PEARCE_IJ a.Ai[om + (size_t)i * n + j] = A[i * ld + j];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        real s = 0;
        for (int k = 0; k < n; ++k) s += A[i * ld + k] * yv[k];
        al[i] = s; a.alpha[ov + i] = s;
    }
    __syncthreads();
    real quad = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) quad += yv[i] * al[i];
    quad = block_sum(quad, red);
    if (threadIdx.x == 0) a.lh[c * a.B + b] = real(-0.5) * ((real)n * real(SVGP_LOG_2PI) + quad + logdet);
    if (a.idx) return;                                            // context likelihood only
    PEARCE_IJ {                                                   // W = Ai K
        real s = 0;
        for (int k = 0; k < n; ++k) s += A[i * ld + k] * K[k * ld + j];
        W[i * ld + j] = s;
    }
    __syncthreads();
    real ce = 0;
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        real pm = 0, kak = 0;
        for (int i = 0; i < n; ++i) { pm += K[t * ld + i] * al[i]; kak += K[t * ld + i] * W[i * ld + t]; }
        const real pv = real(1) - kak;
        const size_t e = (size_t)t * a.B + b;
        const real ep = a.use_rng ? svgp_philox_normal((unsigned long long)a.state[SVGP_ST_RNG_CTR],
                                                       (unsigned long long)(e * 2 + c))
                                  : a.eps_in[c][e];
        a.p_m[c][e] = pm; a.p_v[c][e] = pv; a.eps[c][e] = ep; a.z[c][e] = pm + ep * sqrt(pv);
        const real s2 = a.s2[c][e], p = recip_no_nan(s2), dm = pm - yv[t];
        const real cev = real(0.5) * (real(SVGP_LOG_2PI) + log(s2) + (pv + dm * dm) * p);
        a.row_ce[((size_t)c * a.T + t) * a.B + b] = cev;
        ce += cev;
    }
    ce = block_sum(ce, red);
    if (threadIdx.x == 0) a.ce[c * a.B + b] = ce;
}

// reverse of k_pearce_fwd.  gT = d loss / d (prior-KL term) = -beta / B.
template <int CW>
__global__ __launch_bounds__(256) void k_pearce_bwd(PearceArgs a) {
    extern __shared__ __align__(16) real smem[];
    constexpr int RL = 256 / CW;
    const int b = blockIdx.x, c = blockIdx.y, n = a.n, ld = n + 1;
    const int j = threadIdx.x & (CW - 1), i0 = threadIdx.x / CW;
    real* K = smem;
    real* Ai = K + n * ld;
    real* M = Ai + n * ld;        // K diag(g_pv) K, then Ai M Ai
    real* W = M + n * ld;         // Ai K, then Ai M
    real* al = W + n * ld;        // n
    real* gpm = al + n;
    real* gpv = gpm + n;
    real* u = gpv + n;            // K g_pm
    real* tv = u + n;
    real* red = tv + n;           // 16
    const real gT = svgp_seed_T(0, a.B, a.state), gl = gT * a.seed_lh_scale;
    const real l = *a.ls[c], il2 = real(-0.5) / (l * l), il3 = real(1) / (l * l * l);
    const size_t om = ((size_t)c * a.B + b) * n * n, ov = ((size_t)c * a.B + b) * n;
    const bool full = a.idx == nullptr;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        tv[i] = pearce_time(a, b, i); al[i] = a.alpha[ov + i];
        real g_m = 0, g_v = 0;
        if (full) {
            const size_t e = (size_t)i * a.B + b;
            const real s2 = a.s2[c][e], p = recip_no_nan(s2), zb = a.zbar[c][e], pv = a.p_v[c][e];
            const real gC = gT * (a.tmask ? a.tmask[(size_t)b * a.T + i] : real(1));   // seed of this frame's CE term
            g_v = real(0.5) * gC * p + zb * a.eps[c][e] / (real(2) * sqrt(pv));
            g_m = gC * p * (a.p_m[c][e] - a.y[c][e]) + zb;
        }
        gpm[i] = g_m; gpv[i] = g_v;
    }
    __syncthreads();
    PEARCE_IJ {
        const real d = tv[i] - tv[j];
        K[i * ld + j] = exp(d * d * il2);
        Ai[i * ld + j] = a.Ai[om + (size_t)i * n + j];
    }
    __syncthreads();
    real dl = 0;
    if (full) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            real s = 0;
            for (int k = 0; k < n; ++k) s += K[i * ld + k] * gpm[k];
            u[i] = s;
        }
        PEARCE_IJ {
            real s = 0, w = 0;
            for (int t = 0; t < n; ++t) { s += K[i * ld + t] * gpv[t] * K[t * ld + j]; w += Ai[i * ld + t] * K[t * ld + j]; }
            M[i * ld + j] = s; W[i * ld + j] = w;
        }
        __syncthreads();
        // direct K-bar part folded into the length-scale sum: (g_pm_i alpha_j - 2 (Ai K)_ij g_pv_j) K_ij d_ij^2 / l^3
        PEARCE_IJ {
            const real d = tv[i] - tv[j];
            dl += (gpm[i] * al[j] - real(2) * W[i * ld + j] * gpv[j]) * K[i * ld + j] * d * d * il3;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {       // w = Ai u   (g_pm no longer needed: holds w)
            real s = 0;
            for (int k = 0; k < n; ++k) s += Ai[i * ld + k] * u[k];
            gpm[i] = s;
        }
        PEARCE_IJ {                                               // W = Ai M
            real s = 0;
            for (int t = 0; t < n; ++t) s += Ai[i * ld + t] * M[t * ld + j];
            W[i * ld + j] = s;
        }
        __syncthreads();
        PEARCE_IJ {                                               // M = Ai M Ai
            real s = 0;
            for (int t = 0; t < n; ++t) s += W[i * ld + t] * Ai[t * ld + j];
            M[i * ld + j] = s;
        }
        __syncthreads();
    }
    // Abar_ij = gl/2 (alpha_i alpha_j - Ai_ij) - w_i alpha_j + (Ai M Ai)_ij ;  A = K + diag(var)
    PEARCE_IJ {
        real ab = real(0.5) * gl * (al[i] * al[j] - Ai[i * ld + j]);
        if (full) ab += M[i * ld + j] - gpm[i] * al[j];
        const real d = tv[i] - tv[j];
        dl += ab * K[i * ld + j] * d * d * il3;
        if (i == j) {
            const size_t e = pearce_elem(a, b, i);
            real sb = ab, yb = -gl * al[i];
            if (full) {
                const real s2 = a.s2[c][e], p = recip_no_nan(s2), dm = a.p_m[c][e] - a.y[c][e];
                const real gC = gT * (a.tmask ? a.tmask[(size_t)b * a.T + i] : real(1));
                sb += real(0.5) * gC * (p - (a.p_v[c][e] + dm * dm) * p * p);
                yb += gpm[i] - gC * p * dm;
            }
            if (a.accumulate) { a.s2bar[c][e] += sb; a.ybar[c][e] += yb; }
            else { a.s2bar[c][e] = sb; a.ybar[c][e] = yb; }
        }
    }
    dl = block_sum(dl, red);
    if (threadIdx.x == 0) a.dl_part[c * a.B + b] = dl;
}

// d_ls[c] (+)= sum_b part[c][b]
__global__ void k_pearce_dl(int B, int accumulate, const real* __restrict__ part, real* dl_x, real* dl_y) {
    if (threadIdx.x >= 2) return;
    real s = 0;
    for (int b = 0; b < B; ++b) s += part[threadIdx.x * B + b];
    real* o = threadIdx.x ? dl_y : dl_x;
    *o = accumulate ? *o + s : s;
}

// per-video [elbo, recon, prior_kl, lhood, ce, context lhood, 0] (GPVAE_Pearce_model.py:157-236)
__global__ void k_pearce_assemble(int B, int T, const real* __restrict__ lh, const real* __restrict__ ce,
                                  const real* __restrict__ con_lh, const real* __restrict__ row_recon,
                                  const real* __restrict__ row_ce, const real* __restrict__ tmask,
                                  const real* __restrict__ state, real* __restrict__ o) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    real rec = 0, cev = ce[b] + ce[B + b];
    if (tmask) {       // neural-process ELBO: reconstruction and cross-entropy over the target frames only (:178-182,213-222)
        cev = 0;
        for (int t = 0; t < T; ++t) {
            const real w = tmask[(size_t)b * T + t];
            rec += w * row_recon[(size_t)b * T + t];
            cev += w * (row_ce[(size_t)t * B + b] + row_ce[(size_t)(T + t) * B + b]);
        }
    } else {
        for (int t = 0; t < T; ++t) rec += row_recon[(size_t)b * T + t];
    }
    const real lhood = lh[b] + lh[B + b], clh = con_lh ? con_lh[b] + con_lh[B + b] : real(0);
    const real kl = lhood + cev - clh, beta = state[SVGP_ST_BETA];
    o[b] = rec + beta * kl; o[B + b] = rec; o[2 * B + b] = kl; o[3 * B + b] = lhood; o[4 * B + b] = cev;
    o[5 * B + b] = clh; o[6 * B + b] = 0;
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
    REQ_PTRS(dout, db);
    SVGP_REQUIRE(act == 0 || out != nullptr, SVGP_ERR_INVALID, "activation output is NULL");
    (void)part;
    hipLaunchKernelGGL(k_act_bwd_colsum, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, rows, C, act, out, dout, db);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
extern "C" int svgp_act_bwd_bias_scratch_elems(int C) { (void)C; return 1; }   // the single-launch form needs none

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

// ---- Pearce baseline: exact per-video GP (GPVAE_Pearce_model.py:8-86) ------------------------------------------
// Buffers (caller-owned): per coordinate c in {x, y}, all (T, B): y, s2, p_m, p_v, eps, z, zbar, ybar, s2bar;
// Ai (2,B,n,n), alpha (2,B,n), lh (2,B), ce (2,B), row_ce (2,T,B), dl_part (2,B).  idx (B,n) int32 selects a context set
// (forward: lhood only; reverse: seeds -gT, accumulates into ybar / s2bar).  tmask (B,T) restricts the CE terms.

static int pearce_args(const svgp_pearce_bufs* q, PearceArgs* a) {
    SVGP_REQUIRE(q != nullptr, SVGP_ERR_INVALID, "bufs is NULL");
    SVGP_REQUIRE(q->B >= 1 && q->T >= 1 && q->n >= 1 && q->n <= q->T, SVGP_ERR_INVALID, "bad shape B=%d T=%d n=%d", q->B,
                 q->T, q->n);
    SVGP_REQUIRE(q->n <= 64, SVGP_ERR_UNSUPPORTED, "n=%d: the exact per-video GP keeps its n x n matrices in LDS (n <= 64)",
                 q->n);
    SVGP_REQUIRE(q->idx != nullptr || q->n == q->T, SVGP_ERR_INVALID, "n != T needs an index set");
    REQ_PTRS(q->times, q->ls_x, q->ls_y, q->y_x, q->y_y, q->s2_x, q->s2_y, q->Ai, q->alpha, q->lh);
    memset(a, 0, sizeof(*a));
    a->B = q->B; a->T = q->T; a->n = q->n; a->times = q->times; a->idx = q->idx; a->tmask = q->tmask;
    a->ls[0] = q->ls_x; a->ls[1] = q->ls_y; a->y[0] = q->y_x; a->y[1] = q->y_y; a->s2[0] = q->s2_x; a->s2[1] = q->s2_y;
    a->p_m[0] = q->p_m_x; a->p_m[1] = q->p_m_y; a->p_v[0] = q->p_v_x; a->p_v[1] = q->p_v_y;
    a->eps[0] = q->eps_x; a->eps[1] = q->eps_y; a->z[0] = q->z_x; a->z[1] = q->z_y;
    a->zbar[0] = q->zbar_x; a->zbar[1] = q->zbar_y; a->ybar[0] = q->ybar_x; a->ybar[1] = q->ybar_y;
    a->s2bar[0] = q->s2bar_x; a->s2bar[1] = q->s2bar_y;
    a->Ai = q->Ai; a->alpha = q->alpha; a->lh = q->lh; a->ce = q->ce; a->row_ce = q->row_ce; a->dl_part = q->dl_part;
    return SVGP_OK;
}
static int set_lds(const void* fn, size_t bytes) {
    SVGP_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return SVGP_OK;
}

extern "C" int svgp_pearce_gp_fwd(const svgp_pearce_bufs* q, const double* eps_x, const double* eps_y, const double* state,
                                  void* stream) {
    PearceArgs a;
    int rc = pearce_args(q, &a);
    if (rc) return rc;
    if (!q->idx) {
        REQ_PTRS(q->p_m_x, q->p_m_y, q->p_v_x, q->p_v_y, q->eps_x, q->eps_y, q->z_x, q->z_y, q->ce, q->row_ce, state);
        SVGP_REQUIRE((eps_x == nullptr) == (eps_y == nullptr), SVGP_ERR_INVALID, "give both eps or neither");
    }
    a.eps_in[0] = eps_x; a.eps_in[1] = eps_y; a.use_rng = eps_x == nullptr; a.state = state;
    const size_t lds = ((size_t)3 * a.n * (a.n + 1) + 3 * a.n + 16) * sizeof(real);
    if (a.n <= 32) {
        rc = set_lds((const void*)k_pearce_fwd<32>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_pearce_fwd<32>, dim3(a.B, 2), dim3(256), lds, (hipStream_t)stream, a);
    } else {
        rc = set_lds((const void*)k_pearce_fwd<64>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_pearce_fwd<64>, dim3(a.B, 2), dim3(256), lds, (hipStream_t)stream, a);
    }
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// seed_lh_scale: +1 for the full-data likelihood, -1 for a context likelihood (subtracted in the NP ELBO);
// d_ls_x / d_ls_y: length-scale gradients (accumulate != 0 adds to them and to ybar / s2bar)
extern "C" int svgp_pearce_gp_bwd(const svgp_pearce_bufs* q, double seed_lh_scale, int accumulate, const double* state,
                                  double* d_ls_x, double* d_ls_y, void* stream) {
    PearceArgs a;
    int rc = pearce_args(q, &a);
    if (rc) return rc;
    REQ_PTRS(q->ybar_x, q->ybar_y, q->s2bar_x, q->s2bar_y, q->dl_part, state, d_ls_x, d_ls_y);
    if (!q->idx) REQ_PTRS(q->p_m_x, q->p_m_y, q->p_v_x, q->p_v_y, q->eps_x, q->eps_y, q->zbar_x, q->zbar_y);
    a.state = state; a.seed_lh_scale = seed_lh_scale; a.accumulate = accumulate;
    const size_t lds = ((size_t)4 * a.n * (a.n + 1) + 5 * a.n + 16) * sizeof(real);
    if (a.n <= 32) {
        rc = set_lds((const void*)k_pearce_bwd<32>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_pearce_bwd<32>, dim3(a.B, 2), dim3(256), lds, (hipStream_t)stream, a);
    } else {
        rc = set_lds((const void*)k_pearce_bwd<64>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_pearce_bwd<64>, dim3(a.B, 2), dim3(256), lds, (hipStream_t)stream, a);
    }
    SVGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pearce_dl, dim3(1), dim3(64), 0, (hipStream_t)stream, a.B, accumulate, q->dl_part, d_ls_x, d_ls_y);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// out (7,B) = per-video [elbo, recon, prior_kl, lhood, ce, context lhood, 0]; con_lh (2,B) / tmask (B,T) NULL unless NP
extern "C" int svgp_pearce_elbo_assemble(int B, int T, const double* lh, const double* ce, const double* con_lh,
                                         const double* row_recon, const double* row_ce, const double* tmask,
                                         const double* state, double* out, void* stream) {
    SVGP_REQUIRE(B >= 1 && T >= 1, SVGP_ERR_INVALID, "bad shape");
    REQ_PTRS(lh, ce, row_recon, state, out);
    SVGP_REQUIRE(tmask == nullptr || row_ce != nullptr, SVGP_ERR_INVALID, "a target mask needs the per-frame CE terms");
    hipLaunchKernelGGL(k_pearce_assemble, dim3(nb256(B)), dim3(256), 0, (hipStream_t)stream, B, T, lh, ce, con_lh, row_recon,
                       row_ce, tmask, state, out);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}

// rows of x scaled by w (n rows of C values): dlogits of the context frames are zero in the NP ELBO
__global__ void k_scale_rows(long long tot, int C, const real* __restrict__ w, real* __restrict__ x) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < tot) x[i] *= w[i / C];
}
extern "C" int svgp_scale_rows(long long rows, int C, const double* w, double* x, void* stream) {
    SVGP_REQUIRE(rows >= 1 && C >= 1, SVGP_ERR_INVALID, "bad shape");
    REQ_PTRS(w, x);
    hipLaunchKernelGGL(k_scale_rows, dim3(nb256(rows * C)), dim3(256), 0, (hipStream_t)stream, rows * C, C, w, x);
    SVGP_LAUNCH_CHECK();
    return SVGP_OK;
}
