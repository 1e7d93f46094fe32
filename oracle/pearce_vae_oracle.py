"""BASELINE config 1: `python BALL_experiment.py --elbo VAE` (the reference's CPU-runnable smoke case),
restated for CPU (torch float32 like the reference's ball path, SVGPVAE_model.py:19 / utils.py:138-192).

TEST INFRASTRUCTURE / CPU PLUMBING ONLY - SURVEY 8a row a10: "CPU only; not a kernel target"; parity unpinned
(TF not installable; the reference has no tests).  Restates: build_1d_gp (GPVAE_Pearce_model.py:8-86),
build_pearce_elbo_graphs for type_elbo in {GPVAE_Pearce, VAE} (:89-236), build_MLP_inference_graph /
build_MLP_decoder_graph (VAE_utils.py:9-96), build_video_batch_graph (utils.py:138-192), and the driver
constants of BALL_experiment.py:37-48 (batch 35, tmax 30, 32x32 frames, ball radius 3, model_lt = 0.001 for
--elbo VAE so that the per-video GP prior degenerates to N(0, I)).
"""
import math

import numpy as np
import torch

from .svgpvae_oracle import adam_tf1_step, gauss_cross_entropy


def make_video_batch(tmax=30, px=32, py=32, lt=2.0, batch=35, r=3, generator=None, dtype=torch.float32):
    """utils.py:138-192: ball centre paths ~ GP(0, SE(lt)) via chol(K + 1e-5 I) N(0,1), scaled 0.2 px + 0.5 px."""
    t = torch.arange(tmax, dtype=dtype)
    K = torch.exp(-0.5 * (t[:, None] - t[None, :]) ** 2 / lt ** 2) + 1e-5 * torch.eye(tmax, dtype=dtype)
    paths = torch.linalg.cholesky(K) @ torch.randn(tmax, 2 * batch, dtype=dtype, generator=generator)
    paths = paths.reshape(tmax, batch, 2).permute(1, 0, 2) * 0.2 * px + 0.5 * px
    gx = torch.arange(px, dtype=dtype)[None, None, :, None]
    gy = torch.arange(py, dtype=dtype)[None, None, None, :]
    return (((gx - paths[:, :, 0, None, None]) ** 2 + (gy - paths[:, :, 1, None, None]) ** 2) < r * r).to(dtype)


def init_mlp_params(px=32, py=32, hidden=500, seed=0, dtype=torch.float32):
    """truncated_normal(stddev = 1/sqrt(fan_in)) weights, zero biases (VAE_utils.py:33-35,44-46,80-82,88-90)."""
    g = torch.Generator().manual_seed(seed)

    def tn(i, o):
        w = torch.randn(i, o, dtype=dtype, generator=g).clamp_(-2, 2)
        return w / math.sqrt(i)

    return {"encW1": tn(px * py, hidden), "encB1": torch.zeros(1, hidden, dtype=dtype),
            "encW2": tn(hidden, 4), "encB2": torch.zeros(1, 4, dtype=dtype),
            "decW1": tn(2, hidden), "decB1": torch.zeros(1, hidden, dtype=dtype),
            "decW2": tn(hidden, px * py), "decB2": torch.zeros(1, px * py, dtype=dtype)}


def mlp_inference(p, vid):
    """build_MLP_inference_graph (VAE_utils.py:9-55): (batch,tmax,px,py) -> means (batch,tmax,2), vars = exp(.)."""
    b, t, px, py = vid.shape
    h = torch.tanh(vid.reshape(b * t, px * py) @ p["encW1"] + p["encB1"])
    h = (h @ p["encW2"] + p["encB2"]).reshape(b, t, 4)
    return h[:, :, :2], torch.exp(h[:, :, 2:])


def mlp_decoder(p, z, px, py):
    """build_MLP_decoder_graph (VAE_utils.py:58-96): logits (batch,tmax,px,py)."""
    b, t, _ = z.shape
    h = torch.tanh(z.reshape(b * t, 2) @ p["decW1"] + p["decB1"])
    return (h @ p["decW2"] + p["decB2"]).reshape(b, t, px, py)


def build_1d_gp(X, Y, varY, X_test, lt):
    """GPVAE_Pearce_model.py:8-86 (full_variance=False): posterior mean / variance at X_test and the marginal
    likelihood of each series under an SE(lt) prior with heteroscedastic noise varY."""
    n = X.shape[1]
    ilt = -0.5 / (lt * lt)
    K = torch.exp((X[:, :, None] - X[:, None, :]) ** 2 * ilt) + torch.diag_embed(varY)
    chol = torch.linalg.cholesky(K)
    logdet = 2 * torch.log(torch.diagonal(chol, dim1=1, dim2=2)).sum(1)
    iKY = torch.cholesky_solve(Y[:, :, None], chol)
    quad = (Y[:, None, :] @ iKY).reshape(-1)
    lhood = -0.5 * (n * math.log(2 * math.pi) + quad + logdet)
    Ks = torch.exp((X[:, :, None] - X_test[:, None, :]) ** 2 * ilt)
    p_m = (Ks.transpose(1, 2) @ iKY).squeeze(-1)
    p_v = 1 - (Ks * torch.cholesky_solve(Ks, chol)).sum(1)
    return p_m, p_v, lhood


def pearce_elbo(p, vid, beta, lt, epsilon=None):
    """build_pearce_elbo_graphs (GPVAE_Pearce_model.py:89-236) for type_elbo in {GPVAE_Pearce, VAE}.
    Returns elbo (batch), elbo_recon, elbo_prior_kl, full_p_mu, full_p_var, qnet_mu, qnet_var, pred_vid."""
    b, tmax, px, py = vid.shape
    T = torch.arange(tmax, dtype=vid.dtype).repeat(b, 1)
    qmu, qvar = mlp_inference(p, vid)
    pmx, pvx, lx = build_1d_gp(T, qmu[:, :, 0], qvar[:, :, 0], T, lt)
    pmy, pvy, ly = build_1d_gp(T, qmu[:, :, 1], qvar[:, :, 1], T, lt)
    pmu, pvar = torch.stack([pmx, pmy], 2), torch.stack([pvx, pvy], 2)
    ce = gauss_cross_entropy(pmu, pvar, qmu, qvar).sum((1, 2))
    prior_kl = (lx + ly) - ce
    if epsilon is None:
        epsilon = torch.randn(b, tmax, 2, dtype=vid.dtype)
    logits = mlp_decoder(p, pmu + epsilon * torch.sqrt(pvar), px, py)
    recon = -torch.nn.functional.binary_cross_entropy_with_logits(logits, vid, reduction="none").sum((1, 2, 3))
    return recon + beta * prior_kl, recon, prior_kl, pmu, pvar, qmu, qvar, torch.sigmoid(logits)


def run_ball_vae(steps=20, lr=1e-3, beta0=1.0, seed=0, batch=35, tmax=30):
    """BALL_experiment.py --elbo VAE plumbing: fresh synthetic batch per step (videos are re-synthesised in-graph
    every step, utils.py:138-192), model_lt = 0.001, TF1 Adam on -mean(elbo).  Returns the per-step mean ELBO."""
    g = torch.Generator().manual_seed(seed)
    p = init_mlp_params(seed=seed)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(w) for k, w in p.items()}
    out = []
    for t in range(1, steps + 1):
        vid = make_video_batch(tmax=tmax, batch=batch, lt=2.0, generator=g)
        leaf = {k: w.clone().requires_grad_(True) for k, w in p.items()}
        elbo = pearce_elbo(leaf, vid, beta0, 0.001, epsilon=torch.randn(batch, tmax, 2, generator=g))[0]
        loss = -elbo.mean()
        gs = torch.autograd.grad(loss, list(leaf.values()))
        adam_tf1_step(p, dict(zip(leaf.keys(), gs)), m, v, t, lr)
        out.append(float(elbo.mean()))
    return out
