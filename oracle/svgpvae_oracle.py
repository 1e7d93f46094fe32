"""CPU oracle for the SVGPVAE_Hensman training step (rotated MNIST).

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`svgp-vae_amd/`)
may import this module; only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` do, and only as the checker / the reported
baseline.

PARITY UNPINNED: the reference (ratschlab/SVGP-VAE) is TensorFlow 1.15 +
TensorFlow-Probability 0.8 graph code with no tests, no golden vectors and no
way to execute it in this image (TF is neither installed nor installable).
This file is therefore a float64 restatement written by reading the cited
lines; it is pinned only by (i) closed-form known-answer tests
(`tests/test_oracle_kat.py`), (ii) agreement of its *literal* and *efficient*
formulations, (iii) `torch.autograd.gradcheck`, and (iv) the reference's own
data files used as inputs.  Third-party formulas restated here (not in
/root/reference): TFP 0.8 `psd_kernels.ExpSinSquared`, `Linear`; TF 1.15
`tf.linalg.inv/cholesky`, Keras `Conv2D/Dense/UpSampling2D`, `AdamOptimizer`,
`tf.math.reciprocal_no_nan`, `tf.clip_by_value`.

All citations `file:line` are into /root/reference.
Everything is float64 on CPU (the reference's MNIST path is float64 end to end:
VAE_utils.py:101, SVGPVAE_model.py:404).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

DT = torch.float64
LOG_2PI = 1.8378770664093453  # utils.py:498


# --------------------------------------------------------------------------------------
# small TF primitives
# --------------------------------------------------------------------------------------
def reciprocal_no_nan(x):
    """tf.math.reciprocal_no_nan: 1/x, and 0 where x == 0 (SVGPVAE_model.py:282,330)."""
    safe = torch.where(x == 0, torch.ones_like(x), x)
    return torch.where(x == 0, torch.zeros_like(x), 1.0 / safe)


def add_diagonal_jitter(mat, jitter):
    """SVGPVAE_model.py:13-14."""
    return mat + jitter * torch.eye(mat.shape[-1], dtype=mat.dtype)


def clip_by_value(x, lo, hi):
    """tf.clip_by_value; gradient passes where lo <= x <= hi, else 0 (torch.clamp agrees)."""
    return torch.clamp(x, lo, hi)


def gauss_cross_entropy(mu1, var1, mu2, var2):
    """utils.py:483-504: E_{N(mu1,var1)}[log N(z | mu2, var2)], element-wise."""
    term1 = torch.log(var2)
    term2 = (var1 + mu1 ** 2 - 2 * mu1 * mu2 + mu2 ** 2) / var2
    return -0.5 * (LOG_2PI + term1 + term2)


def KL_term_standard_normal_prior(mean_vector, var_vector):
    """VAE_utils.py:261-272: KL(N(mu, var) || N(0, 1)) summed over everything."""
    return 0.5 * (-torch.sum(torch.log(var_vector)) - mean_vector.numel()
                  + torch.sum(var_vector) + torch.sum(mean_vector ** 2))


# --------------------------------------------------------------------------------------
# GP kernel (TFP restatement) -- SVGPVAE_model.py:416-417, 427-476
# --------------------------------------------------------------------------------------
def exp_sin_squared(x, y, amplitude, length_scale, period=2 * math.pi, diag_only=False):
    """TFP 0.8 psd_kernels.ExpSinSquared on one feature dimension:
    k = a^2 exp(-2 sin^2(pi |x-y| / T) / l^2).  sin^2 is even, so |.| is dropped,
    which also makes the derivative at x == y well defined (0)."""
    d = (x - y) if diag_only else (x[:, None] - y[None, :])
    s = torch.sin(math.pi * d / period)
    return amplitude ** 2 * torch.exp(-2.0 * s * s / length_scale ** 2)


def linear_kernel(xo, yo, normalize=False, diag_only=False):
    """TFP Linear() with all parameters None: <x, y>.  Cosine normalisation as in
    SVGPVAE_model.py:465-474."""
    if diag_only:
        k = torch.sum(xo * yo, dim=1)
        if normalize:
            k = k / (torch.linalg.norm(xo, dim=1) * torch.linalg.norm(yo, dim=1))
        return k
    k = xo @ yo.T
    if normalize:
        k = k * (1.0 / (torch.linalg.norm(xo, dim=1, keepdim=True)
                        @ torch.linalg.norm(yo, dim=1, keepdim=True).T))
    return k


class MnistSVGP:
    """Restatement of `mainSVGP` + `mnistSVGP` (SVGPVAE_model.py:174-476), Hensman and
    Titsias branches, *literal* op sequence (explicit inverses, (b,m,m) lambda tensor,
    full b x b products of which only the diagonal is used)."""

    def __init__(self, titsias, inducing_index_points, object_vectors, l_GP, amplitude,
                 jitter, N_train, K_obj_normalize=False):
        self.titsias = titsias
        self.inducing_index_points = inducing_index_points      # (m, 2+M) [id, angle, o_1..o_M]
        self.object_vectors = object_vectors                    # (n_obj, M) or None
        self.l_GP = l_GP
        self.amplitude = amplitude
        self.jitter = jitter
        self.N_train = N_train
        self.K_obj_normalize = K_obj_normalize

    # SVGPVAE_model.py:427-476
    def kernel_matrix(self, x, y, x_inducing=True, y_inducing=True, diag_only=False):
        x_view, y_view = x[:, 1], y[:, 1]
        if self.object_vectors is None:
            x_object, y_object = x[:, 2:], y[:, 2:]
        else:
            x_object = x[:, 2:] if x_inducing else self.object_vectors[x[:, 0].detach().long()]
            y_object = y[:, 2:] if y_inducing else self.object_vectors[y[:, 0].detach().long()]
        view = exp_sin_squared(x_view, y_view, self.amplitude, self.l_GP, diag_only=diag_only)
        obj = linear_kernel(x_object, y_object, self.K_obj_normalize, diag_only=diag_only)
        return view * obj

    # SVGPVAE_model.py:303-343
    def approximate_posterior_params(self, index_points_test, index_points_train, y, noise):
        b = float(index_points_train.shape[0])
        ip = self.inducing_index_points
        K_mm = self.kernel_matrix(ip, ip)
        K_mm_inv = torch.linalg.inv(add_diagonal_jitter(K_mm, self.jitter))
        K_xx = self.kernel_matrix(index_points_test, index_points_test, False, False, diag_only=True)
        K_xm = self.kernel_matrix(index_points_test, ip, x_inducing=False)
        K_mx = K_xm.T
        K_nm = self.kernel_matrix(index_points_train, ip, x_inducing=False)
        K_mn = K_nm.T
        prec = reciprocal_no_nan(noise)
        sigma_l = K_mm + (self.N_train / b) * (K_mn @ (K_nm * prec[:, None]))
        sigma_l_inv = torch.linalg.inv(add_diagonal_jitter(sigma_l, self.jitter))
        mean_vector = (self.N_train / b) * (K_xm @ (sigma_l_inv @ (K_mn @ (prec * y))))
        K_xm_Sigma_l_K_mx = K_xm @ (sigma_l_inv @ K_mx)
        B = K_xx + torch.diagonal(-(K_xm @ (K_mm_inv @ K_mx)) + K_xm_Sigma_l_K_mx)
        mu_hat = (self.N_train / b) * ((K_mm @ (sigma_l_inv @ K_mn)) @ (prec * y))
        A_hat = K_mm @ (sigma_l_inv @ K_mm)
        return mean_vector, B, mu_hat, A_hat

    # SVGPVAE_model.py:345-378
    def mean_vector_bias_analysis(self, index_points, y, noise):
        b = float(index_points.shape[0])
        ip = self.inducing_index_points
        K_mm = self.kernel_matrix(ip, ip)
        K_bm = self.kernel_matrix(index_points, ip, x_inducing=False)
        K_mb = K_bm.T
        sigma_l = K_mm + (self.N_train / b) * (K_mb @ (torch.diag(reciprocal_no_nan(noise)) @ K_bm))
        sigma_l_inv = torch.linalg.inv(add_diagonal_jitter(sigma_l, self.jitter))
        return (self.N_train / b) * ((K_mm @ (sigma_l_inv @ K_mb)) @ (reciprocal_no_nan(noise) * y))

    # SVGPVAE_model.py:220-301
    def variational_loss(self, x, y, mu_hat, A_hat, noise):
        b = float(x.shape[0])
        ip = self.inducing_index_points
        m = float(ip.shape[0])
        K_mm = self.kernel_matrix(ip, ip)
        K_mm_inv = torch.linalg.inv(add_diagonal_jitter(K_mm, self.jitter))
        K_nn = self.kernel_matrix(x, x, False, False, diag_only=True)
        K_nm = self.kernel_matrix(x, ip, x_inducing=False)
        K_mn = K_nm.T
        if self.titsias:  # :246-259
            cov_mat = torch.diag(noise) + K_nm @ (K_mm_inv @ K_mn)
            trace_term = reciprocal_no_nan(noise) * (K_nn - torch.diagonal(K_nm @ (K_mm_inv @ K_mn)))
            cov_j = add_diagonal_jitter(cov_mat, self.jitter)
            cov_mat_inv = torch.linalg.inv(cov_j)
            cov_mat_chol = torch.linalg.cholesky(cov_j)
            cov_mat_log_det = 2 * torch.sum(torch.log(torch.diagonal(cov_mat_chol)))
            L_2_term = -0.5 * (b * LOG_2PI + cov_mat_log_det + torch.sum(y * (cov_mat_inv @ y))
                               + torch.sum(trace_term))
            return L_2_term, torch.zeros((), dtype=DT)
        # Hensman :261-301
        mean_vector = K_nm @ (K_mm_inv @ mu_hat)
        K_mm_chol = torch.linalg.cholesky(add_diagonal_jitter(K_mm, self.jitter))
        S_chol = torch.linalg.cholesky(add_diagonal_jitter(A_hat, self.jitter))
        K_mm_log_det = 2 * torch.sum(torch.log(torch.diagonal(K_mm_chol)))
        S_log_det = 2 * torch.sum(torch.log(torch.diagonal(S_chol)))
        KL_term = 0.5 * (K_mm_log_det - S_log_det - m + torch.trace(K_mm_inv @ A_hat)
                         + torch.sum(mu_hat * (K_mm_inv @ mu_hat)))
        precision = reciprocal_no_nan(noise)
        K_tilde_terms = precision * (K_nn - torch.diagonal(K_nm @ (K_mm_inv @ K_mn)))
        lambda_mat = K_nm[:, :, None] @ K_nm[:, None, :]                    # (b, m, m)
        lambda_mat = K_mm_inv @ (lambda_mat @ K_mm_inv)                      # (b, m, m)
        trace_terms = precision * torch.diagonal(A_hat @ lambda_mat, dim1=-2, dim2=-1).sum(-1)
        L_3_sum_term = -0.5 * (torch.sum(K_tilde_terms) + torch.sum(trace_terms)
                               + torch.sum(torch.log(noise)) + b * LOG_2PI
                               + torch.sum(precision * (y - mean_vector) ** 2))
        return L_3_sum_term, KL_term


# --------------------------------------------------------------------------------------
# Efficient O(L b m^2 + L m^3) formulation of the same Hensman block
# --------------------------------------------------------------------------------------
def gp_block_efficient(K, Kn, knn, y, s2, jitter, N_train, b_global=None, want_aux=False, kl_form=0):
    """kl_form 1 = the moving-ball SVGP's KL (SVGPVAE_model.py:135-137: A_hat in the place of mu_hat, summed over the
    whole batch of videos = channels here): the last KL summand of channel l becomes L tr(Ki A_l A_l), which has the
    same sum over channels as the reference's per-video scalar sum_l' tr(Ki A_l' A_l').
    All L channels at once.  K (m,m), Kn (b,m), knn (b), y/s2 (b,L) (s2 already clipped).
    Returns p_m (b,L), p_v (b,L), L3 (L,), KL (L,).  Same arithmetic as the literal
    MnistSVGP methods (explicit inverses of K+jI, Sigma+jI; Cholesky only for log-dets)
    but without the (b,m,m) / (b,b) temporaries."""
    b, m = Kn.shape
    L = y.shape[1]
    bg = float(b if b_global is None else b_global)
    c = N_train / bg
    eye = torch.eye(m, dtype=DT)
    Kj = K + jitter * eye
    Ki = torch.linalg.inv(Kj)
    ldK = 2 * torch.sum(torch.log(torch.diagonal(torch.linalg.cholesky(Kj))))
    W = Kn @ Ki                                  # (b,m)
    q = torch.sum(W * Kn, dim=1)                 # (b)
    p = reciprocal_no_nan(s2)                    # (b,L)
    S = torch.einsum('nl,ni,nj->lij', p, Kn, Kn)          # (L,m,m)
    v = torch.einsum('nl,ni->li', p * y, Kn)              # (L,m)
    Sigma = K[None] + c * S
    Si = torch.linalg.inv(Sigma + jitter * eye[None])     # (L,m,m)
    t = torch.einsum('lij,lj->li', Si, v)                 # (L,m)
    p_m = c * (Kn @ t.T)                                  # (b,L)
    r = torch.einsum('ni,lij,nj->nl', Kn, Si, Kn)         # (b,L)
    p_v = knn[:, None] - q[:, None] + r
    mu_hat = c * (t @ K.T)                                # (L,m)
    A_hat = K[None] @ Si @ K[None]                        # (L,m,m)
    u = mu_hat @ Ki.T                                     # (L,m)
    mv = Kn @ u.T                                         # (b,L)
    ldA = 2 * torch.sum(torch.log(torch.diagonal(
        torch.linalg.cholesky(A_hat + jitter * eye[None]), dim1=-2, dim2=-1)), dim=-1)
    last = torch.sum(mu_hat * u, dim=1) if kl_form == 0 else L * torch.einsum('ij,ljk,lki->l', Ki, A_hat, A_hat)
    KL = 0.5 * (ldK - ldA - m + torch.einsum('ij,lji->l', Ki, A_hat) + last)
    Ktil = p * (knn - q)[:, None]
    tr = p * torch.einsum('ni,lij,nj->nl', W, A_hat, W)
    L3 = -0.5 * (Ktil.sum(0) + tr.sum(0) + torch.log(s2).sum(0) + b * LOG_2PI
                 + (p * (y - mv) ** 2).sum(0))
    if want_aux:
        aux = dict(Ki=Ki, ldK=ldK, W=W, q=q, S=S, v=v, Si=Si, t=t, mu_hat=mu_hat, A_hat=A_hat,
                   u=u, mv=mv, ldA=ldA, r=r)
        return p_m, p_v, L3, KL, aux
    return p_m, p_v, L3, KL


def titsias_block_efficient(K, Kn, knn, y, s2, jitter):
    """sum over channels of the Titsias L_2 (SVGPVAE_model.py:246-259) WITHOUT b x b matrices: Woodbury / matrix
    determinant lemma on C = diag(s2 + j) + Kn (K + jI)^-1 Kn^T.  This is the formulation the HIP path implements
    (gp_titsias.hip); tests/test_oracle_kat.py pins it against the literal branch of MnistSVGP.variational_loss."""
    b, m = Kn.shape
    L = y.shape[1]
    eye = torch.eye(m, dtype=DT)
    Kj = K + jitter * eye
    Ki = torch.linalg.inv(Kj)
    ldK = 2 * torch.sum(torch.log(torch.diagonal(torch.linalg.cholesky(Kj))))
    d = s2 + jitter                                       # (b,L)
    S2 = torch.einsum('nl,ni,nj->lij', 1.0 / d, Kn, Kn)
    v2 = torch.einsum('nl,ni->li', y / d, Kn)
    Sig = Kj[None] + S2
    ld2 = 2 * torch.sum(torch.log(torch.diagonal(torch.linalg.cholesky(Sig), dim1=-2, dim2=-1)), dim=-1)
    t2 = torch.einsum('lij,lj->li', torch.linalg.inv(Sig), v2)
    q = torch.sum((Kn @ Ki) * Kn, dim=1)
    rows = (torch.log(d) + y * y / d + reciprocal_no_nan(s2) * (knn - q)[:, None]).sum(0)    # (L,)
    return -0.5 * (b * LOG_2PI + rows + ld2 - ldK - (v2 * t2).sum(1))


# --------------------------------------------------------------------------------------
# mnistVAE (VAE_utils.py:99-162), Keras semantics on NHWC tensors
# --------------------------------------------------------------------------------------
def _conv2d_nhwc(x, w, bias, stride, padding):
    """Keras Conv2D: x (b,H,W,Cin), w (kh,kw,Cin,Cout) [TF layout], padding 'valid'|'same'.
    TF 'SAME': total pad = max((ceil(H/s)-1)*s + k - H, 0), extra on bottom/right."""
    kh, kw = w.shape[0], w.shape[1]
    xin = x.permute(0, 3, 1, 2)
    if padding == 'same':
        H, Wd = x.shape[1], x.shape[2]
        ph = max((math.ceil(H / stride) - 1) * stride + kh - H, 0)
        pw = max((math.ceil(Wd / stride) - 1) * stride + kw - Wd, 0)
        xin = F.pad(xin, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    out = F.conv2d(xin.contiguous(), w.permute(3, 2, 0, 1).contiguous(), bias, stride=stride)
    return out.permute(0, 2, 3, 1)


def _upsample2_nhwc(x):
    """Keras UpSampling2D(size=(2,2)), nearest neighbour."""
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


MNIST_VAE_PARAM_SHAPES = [
    # encoder, VAE_utils.py:114-126
    ("enc_c1_w", (3, 3, 1, 8)), ("enc_c1_b", (8,)),
    ("enc_c2_w", (3, 3, 8, 8)), ("enc_c2_b", (8,)),
    ("enc_c3_w", (3, 3, 8, 8)), ("enc_c3_b", (8,)),
    ("enc_d_w", (32, None)), ("enc_d_b", (None,)),          # Dense(2L): (32, 2L)
    # decoder, VAE_utils.py:128-141
    ("dec_d_w", (None, 128)), ("dec_d_b", (128,)),           # Dense(128): (L, 128)
    ("dec_c1_w", (3, 3, 8, 8)), ("dec_c1_b", (8,)),
    ("dec_c2_w", (3, 3, 8, 8)), ("dec_c2_b", (8,)),
    ("dec_c3_w", (3, 3, 8, 1)), ("dec_c3_b", (1,)),
]


def mnist_vae_param_shapes(L=16):
    out = []
    for name, shp in MNIST_VAE_PARAM_SHAPES:
        if name == "enc_d_w":
            shp = (32, 2 * L)
        elif name == "enc_d_b":
            shp = (2 * L,)
        elif name == "dec_d_w":
            shp = (L, 128)
        out.append((name, shp))
    return out


def glorot_uniform_init(L=16, seed=0):
    """Keras default initialisers: glorot_uniform kernels, zero biases."""
    rng = np.random.RandomState(seed)
    params = {}
    for name, shp in mnist_vae_param_shapes(L):
        if name.endswith("_b"):
            params[name] = np.zeros(shp)
        else:
            if len(shp) == 4:
                rf = shp[0] * shp[1]
                fan_in, fan_out = rf * shp[2], rf * shp[3]
            else:
                fan_in, fan_out = shp
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            params[name] = rng.uniform(-lim, lim, size=shp)
    return params


class MnistVAE:
    """VAE_utils.py:99-162.  `params` is a dict name -> float64 tensor (TF layouts)."""

    def __init__(self, params, L=16):
        self.p = params
        self.L = L

    def encode(self, images):
        p = self.p
        h = F.elu(_conv2d_nhwc(images, p["enc_c1_w"], p["enc_c1_b"], 2, 'valid'))   # 13x13x8
        h = F.elu(_conv2d_nhwc(h, p["enc_c2_w"], p["enc_c2_b"], 2, 'valid'))        # 6x6x8
        h = F.elu(_conv2d_nhwc(h, p["enc_c3_w"], p["enc_c3_b"], 2, 'valid'))        # 2x2x8
        h = h.reshape(h.shape[0], -1)                                              # Flatten (h,w,c)
        enc = h @ p["enc_d_w"] + p["enc_d_b"]
        return enc[:, :self.L], torch.exp(enc[:, self.L:])                         # :151

    def decode(self, z):
        p = self.p
        h = z @ p["dec_d_w"] + p["dec_d_b"]
        h = h.reshape(-1, 4, 4, 8)
        h = F.elu(_conv2d_nhwc(_upsample2_nhwc(h), p["dec_c1_w"], p["dec_c1_b"], 1, 'same'))   # 8x8x8
        h = F.elu(_conv2d_nhwc(_upsample2_nhwc(h), p["dec_c2_w"], p["dec_c2_b"], 1, 'valid'))  # 14x14x8
        h = F.elu(_conv2d_nhwc(_upsample2_nhwc(h), p["dec_c3_w"], p["dec_c3_b"], 1, 'same'))   # 28x28x1
        return h


# --------------------------------------------------------------------------------------
# forward_pass_SVGPVAE (SVGPVAE_model.py:823-936)
# --------------------------------------------------------------------------------------
def forward_pass_SVGPVAE(data_batch, beta, vae, svgp, C_ma, lagrange_mult, alpha, kappa,
                         clipping_qs=False, GECO=False, epsilon=None, formulation="literal",
                         b_global=None, bias_analysis=False):
    """Returns the reference's 16-tuple.  `epsilon` (b,L) is the N(0,1) draw of :901 made an
    explicit input (the reference never seeds TF).  formulation: 'literal' follows the
    reference op by op per channel; 'efficient' uses gp_block_efficient."""
    images, aux_data = data_batch
    _, w, h, c = images.shape
    Kpix = float(w * h * c)
    b = float(images.shape[0])
    qnet_mu, qnet_var = vae.encode(images)
    L = qnet_mu.shape[1]
    if clipping_qs:
        qnet_var = clip_by_value(qnet_var, 1e-3, 10.0)

    if formulation == "literal":
        rec, kl, p_m, p_v = [], [], [], []
        for l in range(L):
            p_m_l, p_v_l, mu_hat_l, A_hat_l = svgp.approximate_posterior_params(
                aux_data, aux_data, qnet_mu[:, l], qnet_var[:, l])
            rec_l, kl_l = svgp.variational_loss(aux_data, qnet_mu[:, l], mu_hat_l, A_hat_l,
                                                noise=qnet_var[:, l])
            rec.append(rec_l); kl.append(kl_l); p_m.append(p_m_l); p_v.append(p_v_l)
        inside_elbo_recon = torch.stack(rec).sum()
        inside_elbo_kl = torch.stack(kl).sum()
        p_m = torch.stack(p_m, dim=1)
        p_v = torch.stack(p_v, dim=1)
    else:
        ip = svgp.inducing_index_points
        K = svgp.kernel_matrix(ip, ip)
        Kn = svgp.kernel_matrix(aux_data, ip, x_inducing=False)
        knn = svgp.kernel_matrix(aux_data, aux_data, False, False, diag_only=True)
        p_m, p_v, L3, KL = gp_block_efficient(K, Kn, knn, qnet_mu, qnet_var, svgp.jitter,
                                              svgp.N_train, b_global=b_global)
        inside_elbo_recon = L3.sum()
        inside_elbo_kl = KL.sum()

    if svgp.titsias:
        inside_elbo = inside_elbo_recon - inside_elbo_kl
    else:
        bg = b if b_global is None else float(b_global)
        inside_elbo = inside_elbo_recon - (bg / svgp.N_train) * inside_elbo_kl

    ce_term = gauss_cross_entropy(p_m, p_v, qnet_mu, qnet_var).sum()
    KL_term = -ce_term + inside_elbo
    if epsilon is None:
        epsilon = torch.randn(p_m.shape, dtype=DT)
    latent_samples = p_m + epsilon * torch.sqrt(p_v)
    recon_images = vae.decode(latent_samples)

    if GECO:  # :908-915
        recon_loss = torch.mean((images - recon_images) ** 2, dim=(1, 2, 3))
        recon_loss = torch.sum(recon_loss - kappa ** 2)
        C_ma = alpha * C_ma + (1 - alpha) * recon_loss / b
        elbo = -KL_term + lagrange_mult * (recon_loss / b + (C_ma - recon_loss / b).detach())
        lagrange_mult = lagrange_mult * torch.exp(C_ma)
    else:      # :917-925
        recon_loss = torch.sum((images - recon_images) ** 2) / Kpix
        elbo = -recon_loss + (beta / float(L)) * KL_term

    if bias_analysis:   # :927-931
        mean_vectors = [svgp.mean_vector_bias_analysis(aux_data, qnet_mu[:, l], qnet_var[:, l]) for l in range(L)]
    else:
        mean_vectors = torch.ones((), dtype=DT)
    return (elbo, recon_loss, KL_term, inside_elbo, ce_term, p_m, p_v, qnet_mu, qnet_var,
            recon_images, inside_elbo_recon, inside_elbo_kl, latent_samples, C_ma, lagrange_mult,
            mean_vectors)


# --------------------------------------------------------------------------------------
# Optimiser: TF1 AdamOptimizer (MNIST_experiment.py:200,207-208)
# --------------------------------------------------------------------------------------
def adam_tf1_step(params, grads, m_state, v_state, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """In-place TF1 Adam on dicts of tensors.  `step` is the 1-based step count t.
    lr_t = lr sqrt(1-b2^t)/(1-b1^t); theta -= lr_t m / (sqrt(v) + eps)  (eps NOT bias-corrected).
    Sparse (IndexedSlices) gradients of gathered tables behave as dense ones with zero rows."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    for k in params:
        g = grads[k]
        m_state[k].mul_(beta1).add_(g, alpha=1 - beta1)
        v_state[k].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        params[k].sub_(lr_t * m_state[k] / (torch.sqrt(v_state[k]) + eps))


# --------------------------------------------------------------------------------------
# One training step / trajectory (MNIST_experiment.py:197-208, 313-355)
# --------------------------------------------------------------------------------------
TRAINABLE_GP = ("inducing_index_points", "l_GP", "amplitude", "object_vectors")


def make_models(params, titsias, jitter, N_train, L, K_obj_normalize=False):
    vae = MnistVAE(params, L=L)
    svgp = MnistSVGP(titsias, params["inducing_index_points"], params.get("object_vectors"),
                     params["l_GP"], params["amplitude"], jitter, N_train, K_obj_normalize)
    return vae, svgp


def loss_and_grads(params, images, aux, epsilon, *, beta, C_ma, lagrange_mult, alpha, kappa,
                   clipping_qs, GECO, jitter, N_train, L, formulation="literal",
                   K_obj_normalize=False, b_global=None, titsias=False):
    """Returns (16-tuple detached, grads dict) of the minimised objective:
    GECO -> `elbo` slot itself, else `-elbo` (MNIST_experiment.py:202-205)."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    vae, svgp = make_models(leaf, titsias, jitter, N_train, L, K_obj_normalize)
    out = forward_pass_SVGPVAE((images, aux), beta, vae, svgp, C_ma, lagrange_mult, alpha, kappa,
                               clipping_qs=clipping_qs, GECO=GECO, epsilon=epsilon,
                               formulation="literal" if titsias else formulation, b_global=b_global)
    objective = out[0] if GECO else -out[0]
    names = list(leaf.keys())
    gs = torch.autograd.grad(objective, [leaf[k] for k in names], allow_unused=True)
    grads = {k: (torch.zeros_like(leaf[k]) if g is None else g) for k, g in zip(names, gs)}
    out = tuple(o.detach() if torch.is_tensor(o) else o for o in out)
    return out, grads


def train_trajectory(params, batches, epsilons, *, beta, lr, alpha_flag, kappa, clipping_qs, GECO,
                     jitter, N_train, L, formulation="literal"):
    """Host state machine of MNIST_experiment.py:313-355: first GECO step uses alpha=0, C_ma and
    lagrange_mult (init 0.0 / 1.0) carried between steps, Adam global step.
    Returns per-step scalars and the final params (params are updated in place on a copy)."""
    params = {k: v.detach().clone() for k, v in params.items()}
    m_state = {k: torch.zeros_like(v) for k, v in params.items()}
    v_state = {k: torch.zeros_like(v) for k, v in params.items()}
    C_ma_ = torch.zeros((), dtype=DT)
    lagr_ = torch.ones((), dtype=DT)
    first_step = True
    log = []
    for t, ((images, aux), eps) in enumerate(zip(batches, epsilons), start=1):
        alpha = 0.0 if (GECO and first_step) else alpha_flag
        out, grads = loss_and_grads(params, images, aux, eps, beta=beta, C_ma=C_ma_,
                                    lagrange_mult=lagr_, alpha=alpha, kappa=kappa,
                                    clipping_qs=clipping_qs, GECO=GECO, jitter=jitter,
                                    N_train=N_train, L=L, formulation=formulation)
        adam_tf1_step(params, grads, m_state, v_state, t, lr)
        if GECO:
            C_ma_, lagr_ = out[13], out[14]
        first_step = False
        log.append(dict(elbo=float(out[0]), recon_loss=float(out[1]), KL_term=float(out[2]),
                        C_ma=float(out[13]), lagrange_mult=float(out[14])))
    return log, params, m_state, v_state


# --------------------------------------------------------------------------------------
# Conditional generation / test path (SVGPVAE_model.py:939-968, 1026-1083; MNIST_experiment.py:457-486)
# --------------------------------------------------------------------------------------
def batching_encode_SVGPVAE(data_batch, vae, clipping_qs=False):
    """SVGPVAE_model.py:939-968."""
    images, aux_data = data_batch
    qnet_mu, qnet_var = vae.encode(images)
    if clipping_qs:
        qnet_var = clip_by_value(qnet_var, 1e-3, 10.0)
    return qnet_mu, qnet_var, aux_data


def bacthing_predict_SVGPVAE_rotated_mnist(test_data_batch, vae, svgp, qnet_mu, qnet_var, aux_data_train,
                                           epsilon=None):
    """SVGPVAE_model.py:1026-1083 (sic spelling).  GP posterior at the test aux data given the encodings
    of the whole train set, sample, decode, per-pixel squared error.  Returns (recon_images, recon_loss)."""
    images_test_batch, aux_data_test_batch = test_data_batch
    _, w, h, _ = images_test_batch.shape
    p_m, p_v = [], []
    for l in range(qnet_mu.shape[1]):
        p_m_l, p_v_l, _, _ = svgp.approximate_posterior_params(aux_data_test_batch, aux_data_train,
                                                               qnet_mu[:, l], qnet_var[:, l])
        p_m.append(p_m_l)
        p_v.append(p_v_l)
    p_m = torch.stack(p_m, dim=1)
    p_v = torch.stack(p_v, dim=1)
    if epsilon is None:
        epsilon = torch.randn(p_m.shape, dtype=DT)
    latent_samples = p_m + epsilon * torch.sqrt(p_v)
    recon = vae.decode(latent_samples)
    recon_loss = torch.sum((images_test_batch - recon) ** 2) / float(w * h)
    return recon, recon_loss
