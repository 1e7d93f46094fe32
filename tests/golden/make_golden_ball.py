"""Small committed fixture for the moving-ball ELBOs (SURVEY 8c "golden vectors to commit"; 8a row a10, 8f rank 3).

    python tests/golden/make_golden_ball.py

Inputs are closed-form functions of the index (`ball_inputs()` is imported by the tests, nothing large is stored); the
expected outputs come from the LITERAL float64 restatement oracle/ball_oracle.py (parity unpinned, see its header).
File: ball_small.npz (a few KB): per ELBO {SVGPVAE_Hensman, SVGPVAE_Titsias, GPVAE_Pearce, VAE, NP} the per-video ELBO
terms, posterior moments and the gradients of the small parameter groups."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ball_oracle as BO  # noqa: E402

DT = torch.float64
CFG = dict(batch=4, T=10, px=8, hidden=12, m=5, beta=0.8, jitter=1e-6, lt=2.0, con_tf=4)
ELBOS = ("SVGPVAE_Hensman", "SVGPVAE_Titsias", "GPVAE_Pearce", "VAE", "NP")
SMALL_GRADS = ("encB1", "encW2", "encB2", "decW1", "decB1")


def ball_inputs():
    c = CFG
    B, T, px, H, m = c["batch"], c["T"], c["px"], c["hidden"], c["m"]
    P = px * px
    b_, t_, i_, j_ = np.meshgrid(np.arange(B), np.arange(T), np.arange(px), np.arange(px), indexing="ij")
    cx, cy = 3.5 + 2.5 * np.sin(0.5 * t_ + b_), 3.5 + 2.5 * np.cos(0.35 * t_ + 2.0 * b_)
    vid = torch.tensor(((i_ - cx) ** 2 + (j_ - cy) ** 2 < 4.0).astype(np.float64))

    def mat(r, c_, s, ph):
        a, b = np.meshgrid(np.arange(r), np.arange(c_), indexing="ij")
        return torch.tensor(s * np.sin(0.37 * a + 0.91 * b + ph + 0.013 * a * b), dtype=DT)

    p = {"encW1": mat(P, H, 1 / np.sqrt(P), 0.1), "encB1": mat(1, H, 0.05, 0.2), "encW2": mat(H, 4, 1 / np.sqrt(H), 0.3),
         "encB2": mat(1, 4, 0.05, 0.4), "decW1": mat(2, H, 0.7, 0.5), "decB1": mat(1, H, 0.05, 0.6),
         "decW2": mat(H, P, 1 / np.sqrt(H), 0.7), "decB2": mat(1, P, 0.05, 0.8)}
    gp = {"ip_x": torch.linspace(1.0, float(T), m, dtype=DT) + 0.1, "l_x": torch.tensor(2.0, dtype=DT),
          "ip_y": torch.linspace(1.0, float(T), m, dtype=DT) - 0.15, "l_y": torch.tensor(2.4, dtype=DT)}
    bb, tt, cc = np.meshgrid(np.arange(B), np.arange(T), np.arange(2), indexing="ij")
    eps = torch.tensor(1.1 * np.cos(0.7 * bb + 1.3 * tt + 2.1 * cc), dtype=DT)
    ran_ind = torch.tensor(np.stack([np.roll(np.arange(T)[::(1 if b % 2 == 0 else -1)], 3 * b) for b in range(B)]).copy())
    return p, gp, vid, eps, ran_ind


def expected(elbo):
    c = CFG
    p, gp, vid, eps, ran_ind = ball_inputs()
    if elbo.startswith("SVGPVAE"):
        out, loss, grads = BO.loss_and_grads({**p, **gp}, vid, eps, beta=c["beta"], titsias="Titsias" in elbo,
                                             jitter=c["jitter"], clipping_qs=True)
        names = ("elbo", "recon", "KL_term", "inside_elbo", "ce_term", "full_p_mu", "full_p_var")
        extra = ("ip_x", "l_x", "ip_y", "l_y")
    else:
        lt = 0.001 if elbo == "VAE" else c["lt"]
        q = {**p, "l_x": torch.tensor(lt, dtype=DT), "l_y": torch.tensor(lt * (1.0 if elbo == "VAE" else 1.2), dtype=DT)}
        out, loss, grads = BO.pearce_loss_and_grads(q, vid, eps, beta=c["beta"], type_elbo=elbo, lt=lt,
                                                    ran_ind=ran_ind if elbo == "NP" else None,
                                                    con_tf=c["con_tf"] if elbo == "NP" else None)
        names = ("elbo", "recon", "prior_kl", "full_p_mu", "full_p_var")
        extra = () if elbo == "VAE" else ("l_x", "l_y")
    d = {f"{elbo}_{n}": out[i].numpy() for i, n in enumerate(names)}
    d[f"{elbo}_loss"] = np.asarray(float(loss))
    for k in SMALL_GRADS + extra:
        d[f"{elbo}_grad_{k}"] = grads[k].numpy()
    return d


if __name__ == "__main__":
    blob = {}
    for e in ELBOS:
        blob.update(expected(e))
    np.savez_compressed(os.path.join(HERE, "ball_small.npz"), **blob)
    print("wrote ball_small.npz:", sum(v.size for v in blob.values()), "values")
