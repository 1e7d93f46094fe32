"""Committed small fixtures (tests/golden/make_golden_small.py): the oracle must keep reproducing them (CPU, through its
OTHER formulation than the one that generated them), and the HIP path must match them (GPU)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import pearce_vae_oracle as P
from oracle import sprites_oracle as SO
from tests.golden import make_golden_small as G

DT = torch.float64
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, c):
    a, c = np.asarray(a, np.float64).reshape(-1), np.asarray(c, np.float64).reshape(-1)
    return float(np.abs(a - c).max() / max(float(np.abs(c).max()), 1e-12))


@pytest.mark.parametrize("K_SE,GECO", [(True, True), (False, False)])
def test_sprites_oracle_efficient_formulation_reproduces_fixture(K_SE, GECO):
    gold = np.load(os.path.join(GOLD, "sprites_small.npz"))
    c = G.SPR
    params, gp, images, ids, eps, seg, rep = G.sprites_inputs()
    kw = dict(beta=0.001, C_ma=torch.tensor(0.02, dtype=DT), lagrange_mult=torch.tensor(1.4, dtype=DT), alpha=0.9,
              kappa=math.sqrt(c["kappa2"]), L=c["L"], L_action=c["La"], jitter=c["jitter"], N_train=c["N_train"],
              segment_ids=seg, repeats=rep, clipping_qs=True, GECO=GECO, K_obj_normalize=not K_SE, K_SE=K_SE)
    out, grads = SO.loss_and_grads(params, gp, (images, ids), eps, formulation="efficient", **kw)
    tag = f"se{int(K_SE)}_geco{int(GECO)}_"
    for name, idx in (("elbo", 0), ("recon_loss", 1), ("KL_term", 2), ("inside_elbo", 3), ("ce_term", 4), ("p_m", 5),
                      ("p_v", 6), ("z", 12)):
        assert _rel(out[idx], gold[tag + name]) < 1e-8, name
    for k in ("inducing_index_points", "GPLVM_action", "enc_d_b", "dec_c7_b", "repr_c3_b", "enc_c1_b"):
        assert _rel(grads[k], gold[tag + "grad_" + k]) < 1e-6, k


def test_pearce_oracle_reproduces_fixture():
    gold = np.load(os.path.join(GOLD, "pearce_small.npz"))
    T, y, var = G.pearce_inputs()
    for lt in (0.001, 2.0):
        p_m, p_v, lh = P.build_1d_gp(T, y, var, T, lt)
        assert _rel(p_m, gold[f"lt{lt}_p_m"]) < 1e-12 and _rel(p_v, gold[f"lt{lt}_p_v"]) < 1e-12
        assert _rel(lh, gold[f"lt{lt}_lhood"]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("K_SE,GECO", [(True, True), (False, False)])
def test_sprites_hip_step_matches_fixture(K_SE, GECO):
    from svgp_vae_amd import sprites as S
    gold = np.load(os.path.join(GOLD, "sprites_small.npz"))
    c = G.SPR
    params, gp, images, ids, eps, seg, rep = G.sprites_inputs()
    svgp = S.spritesSVGP(False, False, gp["inducing_index_points"].numpy(), 'main', c["jitter"], c["N_train"], c["La"],
                         gp["GPLVM_action"].numpy(), c["Lc"], c["L"], K_obj_normalize=not K_SE, K_SE=K_SE)
    init = dict(params)
    init["se"] = torch.stack([gp["l_action"], gp["sigma_action"], gp["l_character"], gp["sigma_character"]])
    eng = S.SpritesStepEngine(S.spritesVAE(c["L"]), S.sprites_representation_network(c["Lc"]), svgp, b_max=c["b"],
                              seg_len=c["frames"], clip_qs=True, geco=GECO, kappa_squared=c["kappa2"], beta=0.001,
                              params=init)
    eng.set_scalars(c_ma=0.02, lagrange=1.4, alpha=0.9)
    dev = eng.dev
    eng.step(images.to(dev), ids.to(dev, DT), eps.to(dev), adam=False)
    got = eng.outputs()
    tag = f"se{int(K_SE)}_geco{int(GECO)}_"
    for name, idx in (("elbo", 0), ("recon_loss", 1), ("KL_term", 2), ("inside_elbo", 3), ("ce_term", 4), ("p_m", 5),
                      ("p_v", 6), ("z", 12)):
        assert _rel(got[idx].cpu().numpy(), gold[tag + name]) < 1e-8, name
    for k in ("inducing_index_points", "GPLVM_action", "enc_d_b", "dec_c7_b", "repr_c3_b", "enc_c1_b"):
        assert _rel(eng.grads[k].cpu().numpy(), gold[tag + "grad_" + k]) < 1e-6, k
    sums = dict(zip(gold[tag + "grad_names"].tolist(), gold[tag + "grad_abs_sums"].tolist()))
    for k, want in sums.items():
        if k in eng.grads:
            assert abs(float(eng.grads[k].abs().sum()) - want) < 1e-6 * max(want, 1e-9), k
