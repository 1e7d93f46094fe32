"""Small committed fixtures for the SPRITES step and the config-1 Pearce VAE (SURVEY 8c "golden vectors to commit").

    python tests/golden/make_golden_small.py

Inputs are closed-form / numpy-RandomState functions of the index (`small_inputs.py` helpers below are imported by the
tests, so nothing large is stored); the expected outputs come from the float64 oracle restatements (parity unpinned, see
oracle/ headers).  Files: sprites_small.npz, pearce_small.npz (a few KB each)."""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pearce_vae_oracle as P  # noqa: E402
from oracle import sprites_oracle as SO  # noqa: E402

DT = torch.float64
SPR = dict(b=8, frames=4, L=6, La=8, Lc=16, m=10, n_act=9, jitter=0.01, N_train=100.0, kappa2=0.0075)


def sprites_inputs():
    c = SPR
    rs = np.random.RandomState(7)
    params = {k: torch.tensor(v, dtype=DT) for k, v in SO.glorot_init(c["L"], c["Lc"], 11).items()}
    for k in params:
        if k.endswith("_b"):
            params[k] = torch.tensor(0.05 * rs.standard_normal(tuple(params[k].shape)), dtype=DT)
    gp = dict(inducing_index_points=torch.tensor(1.5 * rs.standard_normal((c["m"], c["La"] + c["Lc"])), dtype=DT),
              GPLVM_action=torch.tensor(1.5 * rs.standard_normal((c["n_act"], c["La"])), dtype=DT),
              l_action=torch.tensor(5.0, dtype=DT), sigma_action=torch.tensor(1.4, dtype=DT),
              l_character=torch.tensor(7.0, dtype=DT), sigma_character=torch.tensor(1.2, dtype=DT))
    n, y, x, ch = np.meshgrid(np.arange(c["b"]), np.arange(64), np.arange(64), np.arange(3), indexing="ij")
    images = torch.tensor(0.5 + 0.5 * np.sin(0.37 * n + 0.11 * y + 0.23 * x + 1.3 * ch + 0.05 * n * x), dtype=DT)
    ids = torch.tensor(np.arange(c["b"]) * 2 % c["n_act"])
    nn, ll = np.meshgrid(np.arange(c["b"]), np.arange(c["L"]), indexing="ij")
    eps = torch.tensor(np.cos(0.7 * nn + 1.9 * ll) * 1.2, dtype=DT)
    seg, rep = SO.aux_data_sprites_utils(c["b"], c["frames"], c["frames"])
    return params, gp, images, ids, eps, seg, rep


def sprites_expected(K_SE, GECO):
    c = SPR
    params, gp, images, ids, eps, seg, rep = sprites_inputs()
    kw = dict(beta=0.001, C_ma=torch.tensor(0.02, dtype=DT), lagrange_mult=torch.tensor(1.4, dtype=DT), alpha=0.9,
              kappa=math.sqrt(c["kappa2"]), L=c["L"], L_action=c["La"], jitter=c["jitter"], N_train=c["N_train"],
              segment_ids=seg, repeats=rep, clipping_qs=True, GECO=GECO, K_obj_normalize=not K_SE, K_SE=K_SE)
    out, grads = SO.loss_and_grads(params, gp, (images, ids), eps, formulation="literal", **kw)
    return out, grads


def pearce_inputs():
    rs = np.random.RandomState(3)
    b, t = 3, 12
    T = torch.arange(t, dtype=DT).repeat(b, 1)
    y = torch.tensor(rs.standard_normal((b, t)), dtype=DT)
    var = torch.tensor(rs.uniform(0.05, 1.5, (b, t)), dtype=DT)
    return T, y, var


if __name__ == "__main__":
    out = {}
    for K_SE, GECO in ((True, True), (False, False)):
        o, g = sprites_expected(K_SE, GECO)
        tag = f"se{int(K_SE)}_geco{int(GECO)}_"
        for name, idx in (("elbo", 0), ("recon_loss", 1), ("KL_term", 2), ("inside_elbo", 3), ("ce_term", 4), ("p_m", 5), ("p_v", 6),
                          ("z", 12)):
            out[tag + name] = np.asarray(o[idx])
        for k in ("inducing_index_points", "GPLVM_action", "enc_d_b", "dec_c7_b", "repr_c3_b", "enc_c1_b"):
            out[tag + "grad_" + k] = g[k].numpy()
        out[tag + "grad_abs_sums"] = np.array([float(g[k].abs().sum()) for k in sorted(g)])
        out[tag + "grad_names"] = np.array(sorted(g))
    np.savez_compressed(os.path.join(HERE, "sprites_small.npz"), **out)
    T, y, var = pearce_inputs()
    pe = {}
    for lt in (0.001, 2.0):
        p_m, p_v, lh = P.build_1d_gp(T, y, var, T, lt)
        pe[f"lt{lt}_p_m"], pe[f"lt{lt}_p_v"], pe[f"lt{lt}_lhood"] = p_m.numpy(), p_v.numpy(), lh.numpy()
    np.savez_compressed(os.path.join(HERE, "pearce_small.npz"), **pe)
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])
