"""Committed moving-ball fixture (tests/golden/make_golden_ball.py): the oracle must keep reproducing it on CPU -
the SVGPVAE ELBOs through the channel-batched EFFICIENT formulation, not the literal one that generated it - and the
HIP engines must match it on the GPU."""
import os

import numpy as np
import pytest
import torch

from oracle import ball_oracle as BO
from oracle import pearce_vae_oracle as PO
from oracle import svgpvae_oracle as O
from tests.golden import make_golden_ball as G

DT = torch.float64
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ball_small.npz"))


def _rel(a, c):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    a, c = a.reshape(-1), np.asarray(c, np.float64).reshape(-1)
    return float(np.abs(a - c).max() / max(float(np.abs(c).max()), 1e-12))


@pytest.mark.parametrize("elbo", ["SVGPVAE_Hensman", "SVGPVAE_Titsias"])
def test_efficient_formulation_reproduces_svgpvae_fixture(elbo):
    c = G.CFG
    p, gp, vid, eps, _ = G.ball_inputs()
    B, T = c["batch"], c["T"]
    qmu, qvar = PO.mlp_inference(p, vid)
    qvar = torch.clamp(qvar, 1e-6, 1e3)
    times = torch.arange(T, dtype=DT) + 1.0
    inside, pm, pv = torch.zeros(B, dtype=DT), [], []
    for ci, cn in enumerate("xy"):
        z, ls = gp[f"ip_{cn}"], gp[f"l_{cn}"]
        K = BO.se_matrix(z[:, None], z[:, None], ls)
        Kn = BO.se_matrix(times[:, None], z[:, None], ls)
        knn = torch.ones(T, dtype=DT)
        y, s2 = qmu[:, :, ci].T.contiguous(), qvar[:, :, ci].T.contiguous()
        p_m, p_v, L3, KL, aux = O.gp_block_efficient(K, Kn, knn, y, s2, c["jitter"], float(T), want_aux=True, kl_form=1)
        if "Titsias" in elbo:
            inside = inside + O.titsias_block_efficient(K, Kn, knn, y, s2, c["jitter"])
        else:
            klq = torch.einsum('ij,ljk,lki->l', aux["Ki"], aux["A_hat"], aux["A_hat"])
            inside = inside + L3 - (KL - 0.5 * B * klq + 0.5 * klq.sum())
        pm.append(p_m.T); pv.append(p_v.T)
    full_p_mu, full_p_var = torch.stack(pm, 2), torch.stack(pv, 2)
    ce = -O.gauss_cross_entropy(full_p_mu, full_p_var, qmu, qvar).sum((1, 2))
    assert _rel(full_p_mu, GOLD[f"{elbo}_full_p_mu"]) < 1e-8 and _rel(full_p_var, GOLD[f"{elbo}_full_p_var"]) < 1e-8
    assert _rel(inside, GOLD[f"{elbo}_inside_elbo"]) < 1e-8 and _rel(ce, GOLD[f"{elbo}_ce_term"]) < 1e-8
    assert _rel(ce + inside, GOLD[f"{elbo}_KL_term"]) < 1e-8


def test_vae_fixture_is_the_closed_form_limit():
    p, _, vid, _, _ = G.ball_inputs()
    qmu, qvar = PO.mlp_inference(p, vid)
    assert _rel(qmu / (1 + qvar), GOLD["VAE_full_p_mu"]) < 1e-10 and _rel(qvar / (1 + qvar), GOLD["VAE_full_p_var"]) < 1e-10
    kl = -torch.distributions.kl_divergence(torch.distributions.Normal(qmu / (1 + qvar), (qvar / (1 + qvar)).sqrt()),
                                            torch.distributions.Normal(torch.zeros_like(qmu), torch.ones_like(qmu))).sum((1, 2))
    assert _rel(kl, GOLD["VAE_prior_kl"]) < 1e-9


@pytest.mark.parametrize("elbo", G.ELBOS)
def test_oracle_regenerates_fixture(elbo):
    for k, v in G.expected(elbo).items():
        assert _rel(v, GOLD[k]) < 1e-12, k


@pytest.mark.gpu
@pytest.mark.parametrize("elbo", G.ELBOS)
def test_hip_engines_match_fixture(elbo):
    from svgp_vae_amd import ball
    c = G.CFG
    p, gp, vid, eps, ran_ind = G.ball_inputs()
    B, T, px, H, m = c["batch"], c["T"], c["px"], c["hidden"], c["m"]
    flat = {k: (v.reshape(-1) if v.dim() == 2 and v.shape[0] == 1 else v) for k, v in p.items()}
    if elbo.startswith("SVGPVAE"):
        mk = lambda n: ball.SVGP("Titsias" in elbo, m, False, 1, T, 2.0, False, n, c["jitter"], 1, T, 2.0)
        flat.update({k: v.reshape(-1) for k, v in gp.items()})
        eng = ball.BallStepEngine(mk("x"), mk("y"), batch=B, tmax=T, px=px, py=px, hidden=H, clip_qs=True, beta=c["beta"],
                                  params=flat)
        eng.step(vid.cuda(), eps.cuda(), adam=False)
        out = eng.outputs()
        got = dict(elbo=out[0], recon=out[1], KL_term=out[2], inside_elbo=out[3], ce_term=out[4], full_p_mu=out[5],
                   full_p_var=out[6])
    else:
        lt = 0.001 if elbo == "VAE" else c["lt"]
        flat.update(l_x=torch.tensor([lt], dtype=DT), l_y=torch.tensor([lt * (1.0 if elbo == "VAE" else 1.2)], dtype=DT))
        eng = ball.PearceStepEngine(elbo, lt, 0.5, elbo != "VAE", 2.0, batch=B, tmax=T, px=px, py=px, hidden=H,
                                    beta=c["beta"], params=flat)
        eng.step(vid.cuda(), eps.cuda(), adam=False, ran_ind=ran_ind.numpy() if elbo == "NP" else None,
                 con_tf=c["con_tf"] if elbo == "NP" else None)
        out = eng.outputs()
        got = dict(elbo=out[0], recon=out[1], prior_kl=out[2], full_p_mu=out[3], full_p_var=out[4])
    for k, v in got.items():
        assert _rel(v, GOLD[f"{elbo}_{k}"]) < 1e-8, k
    assert abs(-eng.scalars()["elbo"] - float(GOLD[f"{elbo}_loss"])) < 1e-9 * abs(float(GOLD[f"{elbo}_loss"]))
    eng.stream.synchronize()
    for key in GOLD.files:
        if key.startswith(f"{elbo}_grad_"):
            name = key[len(elbo) + 6:]
            assert _rel(eng.grads[name], GOLD[key]) < 1e-7, name
