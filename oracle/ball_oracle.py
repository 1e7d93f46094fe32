"""Moving-ball SVGP-VAE (`BALL_experiment.py --elbo SVGPVAE_Hensman | SVGPVAE_Titsias`), restated for CPU.

TEST INFRASTRUCTURE ONLY - imported by tests/ and tests/golden/make_golden_ball.py, never by the product
path.  Parity unpinned: TensorFlow 1.15 / TFP 0.8 cannot be installed here and the reference holds no tests
or golden vectors for this path; what pins it are the identities of tests/test_oracle_kat.py (the literal
per-video form below == the channel-batched efficient form the HIP kernels implement) and autograd.

Literal restatement (same op sequence, same shapes, same quirks) of
  SVGP.__init__ / variational_loss / approximate_posterior_params      SVGPVAE_model.py:17-171
  build_SVGPVAE_elbo_graph                                               SVGPVAE_model.py:638-715
  the loss / optimiser of BALL_experiment.py:116-136 (loss = -mean(elbo), TF1 Adam, optional gradient clip)
  Make_Video_batch semantics through pearce_vae_oracle.make_video_batch  utils.py:138-192
Quirk kept: the Hensman KL of the ball SVGP uses A_hat where mu_hat is meant and reduces over the WHOLE batch
(SVGPVAE_model.py:135-137): every video's KL carries  1/2 sum_{b,i} A_hat[b,i,:]^T K_mm^-1 A_hat[b,i,:].
"""
import math

import torch

from .pearce_vae_oracle import mlp_decoder, mlp_inference
from .svgpvae_oracle import adam_tf1_step, gauss_cross_entropy, reciprocal_no_nan

DT = torch.float64
LOG_2PI = math.log(2 * math.pi)


def se_matrix(x, y, length_scale):
    """tfk.ExponentiatedQuadratic(amplitude=None, length_scale).matrix on 1-feature inputs (SVGPVAE_model.py:60):
    x (..., n, 1), y (..., k, 1) -> (..., n, k)."""
    d = x[..., :, None, 0] - y[..., None, :, 0]
    return torch.exp(-0.5 * d * d / (length_scale * length_scale))


class BallSVGP:
    """SVGP (SVGPVAE_model.py:17-171).  Parameters are plain tensors so that autograd can reach them."""

    def __init__(self, titsias, inducing_index_points, l_GP, jitter):
        self.titsias, self.jitter = titsias, jitter
        self.inducing_index_points = inducing_index_points      # (m)
        self.l_GP = l_GP                                        # scalar tensor

    @staticmethod
    def initial_inducing_points(num_inducing_points, fixed_inducing_points, tmin, tmax, ip_min, ip_max):
        """:45-51"""
        lo, hi = (tmin, tmax) if fixed_inducing_points else (ip_min, ip_max)
        return torch.linspace(float(lo), float(hi), num_inducing_points, dtype=DT)

    def _mats(self, x):
        z = self.inducing_index_points
        eye = torch.eye(z.shape[0], dtype=x.dtype)
        K_mm = se_matrix(z[:, None], z[:, None], self.l_GP)                      # (m,m)
        K_mm_inv = torch.linalg.inv(K_mm + self.jitter * eye)
        K_nn = se_matrix(x[:, :, None], x[:, :, None], self.l_GP)               # (batch,T,T)
        K_nm = se_matrix(x[:, :, None], z[None, :, None], self.l_GP)            # (batch,T,m)
        return K_mm, K_mm_inv, K_nn, K_nm, K_nm.transpose(1, 2), eye

    def variational_loss(self, x, y, noise, mu_hat, A_hat):
        """:62-140.  Returns (sum_term (batch), KL_term (batch) or 0.0)."""
        T = float(x.shape[1])
        m = float(self.inducing_index_points.shape[0])
        precision = reciprocal_no_nan(noise)
        K_mm, K_mm_inv, K_nn, K_nm, K_mn, eye = self._mats(x)
        if self.titsias:
            eyeT = torch.eye(x.shape[1], dtype=x.dtype)
            low = K_nm @ (K_mm_inv @ K_mn)
            cov_mat = torch.diag_embed(noise) + low
            cov_mat_inv = torch.linalg.inv(cov_mat + self.jitter * eyeT)
            chol = torch.linalg.cholesky(cov_mat + self.jitter * eyeT)
            log_det = 2 * torch.log(torch.diagonal(chol, dim1=1, dim2=2)).sum(1)
            trace_term = precision * torch.diagonal(K_nn - low, dim1=1, dim2=2)
            L2 = -0.5 * (T * LOG_2PI + log_det + (y * (cov_mat_inv @ y[:, :, None])[:, :, 0]).sum(1) + trace_term.sum(1))
            return L2, 0.0
        mean_vector = (K_nm @ (K_mm_inv @ mu_hat[:, :, None]))[:, :, 0]
        K_tilde = precision * torch.diagonal(K_nn - K_nm @ (K_mm_inv @ K_mn), dim1=1, dim2=2)
        lam = K_nm[:, :, :, None] @ K_nm[:, :, None, :]                          # (batch,T,m,m)
        lam = K_mm_inv @ (lam @ K_mm_inv)
        trace_terms = precision * torch.diagonal(A_hat[:, None] @ lam, dim1=2, dim2=3).sum(2)
        L3 = -0.5 * (K_tilde.sum(1) + trace_terms.sum(1) + torch.log(noise).sum(1) + T * LOG_2PI +
                     (precision * (y - mean_vector) ** 2).sum(1))
        ld_K = 2 * torch.log(torch.diagonal(torch.linalg.cholesky(K_mm + self.jitter * eye))).sum()
        ld_S = 2 * torch.log(torch.diagonal(torch.linalg.cholesky(A_hat + self.jitter * eye), dim1=1, dim2=2)).sum(1)
        # :135-137 - A_hat in the place of mu_hat, reduce_sum over every axis (scalar broadcast to all videos)
        quirk = (A_hat * (A_hat @ K_mm_inv.T)).sum()
        KL = 0.5 * (ld_K - ld_S - m + torch.diagonal(K_mm_inv @ A_hat, dim1=1, dim2=2).sum(1) + quirk)
        return L3, KL

    def approximate_posterior_params(self, index_points, y, noise):
        """:142-171.  Returns mean (batch,T), B (batch,T,T), mu_hat (batch,m), A_hat (batch,m,m)."""
        K_mm, K_mm_inv, K_nn, K_nm, K_mn, eye = self._mats(index_points)
        p = reciprocal_no_nan(noise)
        sigma_l = K_mm + K_mn @ (torch.diag_embed(p) @ K_nm)
        sigma_l_inv = torch.linalg.inv(sigma_l + self.jitter * eye)
        KSK = K_nm @ (sigma_l_inv @ K_mn)
        mean_vector = (KSK @ (p * y)[:, :, None])[:, :, 0]
        B = K_nn - K_nm @ (K_mm_inv @ K_mn) + KSK
        mu_hat = ((K_mm @ (sigma_l_inv @ K_mn)) @ (p * y)[:, :, None])[:, :, 0]
        A_hat = K_mm @ (sigma_l_inv @ K_mm)
        return mean_vector, B, mu_hat, A_hat


def build_SVGPVAE_elbo_graph(p, vid_batch, beta, svgp_x, svgp_y, clipping_qs=False, epsilon=None):
    """SVGPVAE_model.py:638-715.  `p` = MLP parameters (pearce_vae_oracle.init_mlp_params names).
    Returns the reference's tuple up to (not including) globals()."""
    batch, tmax, px, py = vid_batch.shape
    T = torch.arange(tmax, dtype=vid_batch.dtype) + 1.0
    batch_T = T.repeat(batch, 1)
    qnet_mu, qnet_var = mlp_inference(p, vid_batch)
    if clipping_qs:
        qnet_var = torch.clamp(qnet_var, 1e-6, 1e3)
    p_m_x, p_v_x, mu_hat_x, A_hat_x = svgp_x.approximate_posterior_params(batch_T, qnet_mu[:, :, 0], qnet_var[:, :, 0])
    p_m_y, p_v_y, mu_hat_y, A_hat_y = svgp_y.approximate_posterior_params(batch_T, qnet_mu[:, :, 1], qnet_var[:, :, 1])
    rx, kx = svgp_x.variational_loss(batch_T, qnet_mu[:, :, 0], qnet_var[:, :, 0], mu_hat_x, A_hat_x)
    ry, ky = svgp_y.variational_loss(batch_T, qnet_mu[:, :, 1], qnet_var[:, :, 1], mu_hat_y, A_hat_y)
    inside_elbo_recon = rx + ry
    inside_elbo_kl = kx + ky
    inside_elbo = inside_elbo_recon - inside_elbo_kl
    cov_mean_x, cov_mean_y = p_v_x.mean(0), p_v_y.mean(0)
    full_p_mu = torch.stack([p_m_x, p_m_y], 2)
    full_p_var = torch.stack([torch.diagonal(p_v_x, dim1=1, dim2=2), torch.diagonal(p_v_y, dim1=1, dim2=2)], 2)
    ce_term = -gauss_cross_entropy(full_p_mu, full_p_var, qnet_mu, qnet_var).sum((1, 2))
    if epsilon is None:
        epsilon = torch.randn(batch, tmax, 2, dtype=vid_batch.dtype)
    latent_samples = full_p_mu + epsilon * torch.sqrt(torch.clamp(full_p_var, 1e-4, 1000))
    logits = mlp_decoder(p, latent_samples, px, py)
    pred_vid = torch.sigmoid(logits)
    recon_term = -torch.nn.functional.binary_cross_entropy_with_logits(logits, vid_batch, reduction="none").sum((1, 2, 3))
    KL_term = ce_term + inside_elbo
    elbo = recon_term + beta * KL_term
    return (elbo, recon_term, KL_term, inside_elbo, ce_term, full_p_mu, full_p_var, qnet_mu, qnet_var, pred_vid,
            svgp_x.l_GP, svgp_y.l_GP, inside_elbo_recon, inside_elbo_kl, svgp_x.inducing_index_points,
            svgp_y.inducing_index_points, cov_mean_x, cov_mean_y)


PARAM_ORDER = ("encW1", "encB1", "encW2", "encB2", "decW1", "decB1", "decW2", "decB2", "ip_x", "l_x", "ip_y", "l_y")


def loss_and_grads(params, vid_batch, epsilon, *, beta, titsias, jitter, clipping_qs=False):
    """loss = -mean(elbo) (BALL_experiment.py:117) and its gradient wrt every entry of `params`
    (MLP weights + ip_x, l_x, ip_y, l_y).  Returns (outputs tuple, loss, grads dict)."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    sx = BallSVGP(titsias, leaf["ip_x"], leaf["l_x"], jitter)
    sy = BallSVGP(titsias, leaf["ip_y"], leaf["l_y"], jitter)
    out = build_SVGPVAE_elbo_graph(leaf, vid_batch, beta, sx, sy, clipping_qs, epsilon)
    loss = -out[0].mean()
    gs = torch.autograd.grad(loss, [leaf[k] for k in PARAM_ORDER], allow_unused=True)
    grads = {k: (torch.zeros_like(leaf[k]) if g is None else g) for k, g in zip(PARAM_ORDER, gs)}
    return tuple(o.detach() if torch.is_tensor(o) else o for o in out), loss.detach(), grads


def train_trajectory(params, vids, epsilons, *, beta, titsias, jitter, clipping_qs=False, lr=1e-3, clip_grad=False,
                     train_ip=True, train_gp=True):
    """BALL_experiment.py:120-136,213-217: TF1 Adam (default lr 1e-3) on -mean(elbo), optional element-wise
    gradient clip to +-1e5; fixed inducing points / GP parameters are constants (not in train_vars)."""
    p = {k: v.clone() for k, v in params.items()}
    ms = {k: torch.zeros_like(v) for k, v in p.items()}
    vs = {k: torch.zeros_like(v) for k, v in p.items()}
    elbos = []
    for t, (vid, eps) in enumerate(zip(vids, epsilons), 1):
        out, loss, g = loss_and_grads(p, vid, eps, beta=beta, titsias=titsias, jitter=jitter, clipping_qs=clipping_qs)
        if clip_grad:
            g = {k: torch.clamp(v, -100000.0, 100000.0) for k, v in g.items()}
        frozen = ([] if train_ip else ["ip_x", "ip_y"]) + ([] if train_gp else ["l_x", "l_y"])
        keys = [k for k in p if k not in frozen]
        adam_tf1_step({k: p[k] for k in keys}, {k: g[k] for k in keys}, {k: ms[k] for k in keys},
                      {k: vs[k] for k in keys}, t, lr)
        elbos.append(float(out[0].mean()))
    return p, elbos


# ---------------------------------------------------------------------------------------------------------
# Pearce baseline with trainable length scales and the neural-process ELBO (GPVAE_Pearce_model.py:89-236)
# ---------------------------------------------------------------------------------------------------------
def pearce_elbo_graphs(p, vid_batch, beta, type_elbo, l_x, l_y, lt, epsilon, ran_ind=None, con_tf=None):
    """build_pearce_elbo_graphs for type_elbo in {GPVAE_Pearce, VAE, NP}.  l_x / l_y: the length scales of the two
    full-data GPs (trainable under GP_joint, else equal to lt); lt: the constant model length scale the NP context
    likelihoods use (:152-153).  ran_ind (batch,tmax) long permutations, con_tf = number of context frames (:121-137;
    injected instead of drawn).  Returns (elbo, elbo_recon, elbo_prior_kl, full_p_mu, full_p_var, qnet_mu, qnet_var, pred)."""
    from .pearce_vae_oracle import build_1d_gp
    batch, tmax, px, py = vid_batch.shape
    T = torch.arange(tmax, dtype=vid_batch.dtype)
    batch_T = T.repeat(batch, 1)
    qnet_mu, qnet_var = mlp_inference(p, vid_batch)
    if type_elbo == "NP":
        con_ind, tar_ind = ran_ind[:, :con_tf], ran_ind[:, con_tf:]
        con_T = T[con_ind]
        con_lm = torch.gather(qnet_mu, 1, con_ind[:, :, None].expand(-1, -1, 2))
        con_lv = torch.gather(qnet_var, 1, con_ind[:, :, None].expand(-1, -1, 2))
        con_lhood = build_1d_gp(con_T, con_lm[:, :, 0], con_lv[:, :, 0], batch_T, lt)[2] + \
            build_1d_gp(con_T, con_lm[:, :, 1], con_lv[:, :, 1], batch_T, lt)[2]
    p_mx, p_vx, lhx = build_1d_gp(batch_T, qnet_mu[:, :, 0], qnet_var[:, :, 0], batch_T, l_x)
    p_my, p_vy, lhy = build_1d_gp(batch_T, qnet_mu[:, :, 1], qnet_var[:, :, 1], batch_T, l_y)
    full_p_mu, full_p_var = torch.stack([p_mx, p_my], 2), torch.stack([p_vx, p_vy], 2)
    full_lhood = lhx + lhy
    sin_ce = gauss_cross_entropy(full_p_mu, full_p_var, qnet_mu, qnet_var).sum(2)          # (batch,tmax)
    if epsilon is None:
        epsilon = torch.randn(batch, tmax, 2, dtype=vid_batch.dtype)
    logits = mlp_decoder(p, full_p_mu + epsilon * torch.sqrt(full_p_var), px, py)
    sin_recon = -torch.nn.functional.binary_cross_entropy_with_logits(logits, vid_batch, reduction="none").sum((2, 3))
    if type_elbo == "NP":
        prior_kl = full_lhood - torch.gather(sin_ce, 1, tar_ind).sum(1) - con_lhood
        recon = torch.gather(sin_recon, 1, tar_ind).sum(1)
    else:
        prior_kl = full_lhood - sin_ce.sum(1)
        recon = sin_recon.sum(1)
    return recon + beta * prior_kl, recon, prior_kl, full_p_mu, full_p_var, qnet_mu, qnet_var, torch.sigmoid(logits)


PEARCE_PARAM_ORDER = ("encW1", "encB1", "encW2", "encB2", "decW1", "decB1", "decW2", "decB2", "l_x", "l_y")


def pearce_loss_and_grads(params, vid_batch, epsilon, *, beta, type_elbo, lt, ran_ind=None, con_tf=None):
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    out = pearce_elbo_graphs(leaf, vid_batch, beta, type_elbo, leaf["l_x"], leaf["l_y"], lt, epsilon, ran_ind, con_tf)
    loss = -out[0].mean()
    gs = torch.autograd.grad(loss, [leaf[k] for k in PEARCE_PARAM_ORDER], allow_unused=True)
    grads = {k: (torch.zeros_like(leaf[k]) if g is None else g) for k, g in zip(PEARCE_PARAM_ORDER, gs)}
    return tuple(o.detach() for o in out), loss.detach(), grads
