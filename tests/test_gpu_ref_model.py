"""The HIP step against the REFERENCE'S OWN model code executed on the functional TensorFlow stand-in
(tests/golden/ref_model_cfg2.npz, written by tests/golden/make_ref_model_fixtures.py from /root/reference/SVGPVAE_model.py,
VAE_utils.py, utils.py in the build container): the 16-tuple of forward_pass_SVGPVAE and the gradients of the minimised
objective (MNIST_experiment.py:202-205) at BASELINE configs[1] (b = 256, m = 32, L = 16) under GECO and the beta-ELBO, and at the
ragged batch b = 210 with K_obj_normalize and without clipping.  Bars: scalars 1e-9, row quantities 1e-8, gradients 1e-7 (the
stand-in restates TensorFlow's arithmetic: this checks the op sequence, "parity" stays "partial")."""
import math
import os

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
DT = torch.float64
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag,geco,clip,norm", [("geco", True, True, False), ("beta", False, True, False),
                                                ("geco_norm_ragged", True, False, True)])
def test_hip_step_equals_the_reference_code_on_the_stand_in(golden, tag, geco, clip, norm):
    ref = dict(np.load(os.path.join(G, "ref_model_cfg2.npz")))
    lo, hi = (int(v) for v in ref[tag + "__rows"])
    params, images, aux, eps = H.golden_problem(golden, rows=slice(lo, hi))
    b = hi - lo
    eng = H.engine_for(params, b, geco=geco, clip_qs=clip, K_obj_normalize=norm, N_train=4050.0, jitter=1e-6, beta=0.001,
                       alpha=0.9, kappa_squared=0.02)
    eng.set_scalars(c_ma=0.013, lagrange=1.7, alpha=0.9)          # the state forward_pass_SVGPVAE was called with
    dev = eng.device
    eng.bind(images.to(dev), aux.to(dev), eps.to(dev))
    eng.run(adam=False)
    eng.synchronize()
    sc = eng.scalars()
    for k, n in (("elbo", "elbo"), ("recon_loss", "recon_loss"), ("kl_term", "KL_term"), ("inside_elbo", "inside_elbo"),
                 ("ce_term", "ce_term"), ("inside_recon", "inside_elbo_recon"), ("inside_kl", "inside_elbo_kl")):
        want = float(ref[f"{tag}__{n}"])
        assert abs(sc[k] - want) <= 1e-9 * max(1.0, abs(want)), (k, sc[k], want)
    if geco:
        assert abs(sc["c_ma"] - float(ref[tag + "__C_ma"])) <= 1e-9 and abs(sc["lagrange"] - float(ref[tag + "__lagrange_mult"])) <= 1e-9
    L = 16
    for field, n in (("p_m", "p_m"), ("p_v", "p_v"), ("qnet_mu", "qnet_mu"), ("qnet_var", "qnet_var"), ("z", "latent_samples")):
        assert H.relerr(eng.ws_view(field, (b, L)), ref[f"{tag}__{n}"]) < 1e-8, n
    if f"{tag}__recon_images" in ref:
        assert H.relerr(eng.ws_view("recon", (b, 784)), ref[f"{tag}__recon_images"].reshape(b, 784)) < 1e-8
    g = eng.grads()
    for k in g:
        assert H.relerr(g[k], ref[f"{tag}__grad__{k}"]) < 1e-7, (k, H.relerr(g[k], ref[f"{tag}__grad__{k}"]))
