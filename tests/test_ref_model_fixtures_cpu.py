"""oracle/svgpvae_oracle.py against the REFERENCE'S OWN model code executed on a functional TensorFlow stand-in
(tests/golden/make_ref_model_fixtures.py -> tests/golden/ref_model_cfg2.npz).  The generator imports
/root/reference/SVGPVAE_model.py, VAE_utils.py and utils.py in the build container and runs mnistSVGP.kernel_matrix,
mainSVGP.approximate_posterior_params / variational_loss / mean_vector_bias_analysis, mnistVAE.encode / decode and
forward_pass_SVGPVAE as the reference wrote them, every `tf.*` call mapped one to one onto float64 torch; gradients by autograd
through that code.  This pins the OP SEQUENCE of the hot path -- a transcription slip in the oracle shows up here -- but not
TensorFlow's own arithmetic, which the stand-in restates: "parity" stays "partial" (DESIGN.md section 2).

Cases: GECO and beta-ELBO at b = 256 with clip_qs; GECO at the ragged b = 210 with K_obj_normalize and without clipping; the
Titsias branch; bias_analysis; the three argument patterns of kernel_matrix; posterior parameters at other test points."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import svgpvae_oracle as O
from tests import helpers as H

DT = torch.float64
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ("elbo", "recon_loss", "KL_term", "inside_elbo", "ce_term", "p_m", "p_v", "qnet_mu", "qnet_var", "recon_images",
         "inside_elbo_recon", "inside_elbo_kl", "latent_samples", "C_ma", "lagrange_mult")


@pytest.fixture(scope="module")
def ref():
    return dict(np.load(os.path.join(G, "ref_model_cfg2.npz")))


def _inputs(golden, rows):
    gin, _ = golden
    params = {k[4:]: torch.tensor(v, dtype=DT) for k, v in gin.items() if k.startswith("vae_")}
    for k in ("inducing_index_points", "l_GP", "amplitude", "object_vectors"):
        params[k] = torch.tensor(gin[k], dtype=DT)
    sl = slice(int(rows[0]), int(rows[1]))
    return params, tuple(torch.tensor(gin[k][sl], dtype=DT) for k in ("images", "aux", "epsilon"))


@pytest.mark.parametrize("tag,kw,formulation,tol", [
    ("geco", dict(GECO=True, clipping_qs=True), "literal", 1e-10),
    ("geco", dict(GECO=True, clipping_qs=True), "efficient", 1e-8),
    ("beta", dict(GECO=False, clipping_qs=True), "literal", 1e-10),
    ("geco_norm_ragged", dict(GECO=True, clipping_qs=False, K_obj_normalize=True), "literal", 1e-10),
    ("geco_norm_ragged", dict(GECO=True, clipping_qs=False, K_obj_normalize=True), "efficient", 1e-8),
    ("titsias", dict(GECO=False, clipping_qs=True, titsias=True), "literal", 1e-10)])
def test_oracle_step_equals_the_reference_code_on_the_stand_in(golden, ref, tag, kw, formulation, tol):
    params, (images, aux, eps) = _inputs(golden, ref[tag + "__rows"])
    out, grads = O.loss_and_grads(params, images, aux, eps, beta=0.001, C_ma=torch.tensor(0.013, dtype=DT),
                                  lagrange_mult=torch.tensor(1.7, dtype=DT), alpha=0.9, kappa=math.sqrt(0.02), jitter=1e-6,
                                  N_train=4050.0, L=16, formulation=formulation, **kw)
    for i, n in enumerate(NAMES):
        if f"{tag}__{n}" not in ref:
            continue
        want = torch.tensor(ref[f"{tag}__{n}"], dtype=DT)
        got = torch.as_tensor(out[i], dtype=DT)
        scale = max(1.0, float(want.abs().max()))
        assert float((got.reshape(-1) - want.reshape(-1)).abs().max()) <= tol * scale, (n, float((got.reshape(-1) - want.reshape(-1)).abs().max()))
    for k, g in grads.items():
        want = torch.tensor(ref[f"{tag}__grad__{k}"], dtype=DT)
        # gradient tolerance: the literal form repeats the reference's operations (1e-9); the efficient form reorders them
        assert H.relerr(g, want) < (1e-9 if formulation == "literal" else 1e-6), (k, H.relerr(g, want))


def test_bias_analysis_mean_vectors(golden, ref):
    params, (images, aux, eps) = _inputs(golden, ref["bias__rows"])
    vae, svgp = O.make_models(params, False, 1e-6, 4050.0, 16)
    out = O.forward_pass_SVGPVAE((images, aux), 0.001, vae, svgp, torch.tensor(0.013, dtype=DT), torch.tensor(1.7, dtype=DT), 0.9,
                                 math.sqrt(0.02), clipping_qs=True, GECO=True, epsilon=eps, bias_analysis=True)
    got = torch.stack(list(out[15]))
    assert H.relerr(got, torch.tensor(ref["bias__mean_vectors"], dtype=DT)) < 1e-10


def test_kernel_matrix_argument_patterns_and_posterior_at_other_points(golden, ref):
    params, (images, aux, eps) = _inputs(golden, (0, 64))
    gin, _ = golden
    vae, svgp = O.make_models(params, False, 1e-6, 4050.0, 16)
    ip = params["inducing_index_points"]
    assert H.relerr(svgp.kernel_matrix(ip, ip), ref["km__K_mm"]) < 1e-13
    assert H.relerr(svgp.kernel_matrix(aux, ip, x_inducing=False), ref["km__K_nm"]) < 1e-13
    assert H.relerr(svgp.kernel_matrix(aux, aux, False, False, diag_only=True), ref["km__K_nn_diag"]) < 1e-13
    test_aux = torch.tensor(gin["aux"][300:340], dtype=DT)
    y, noise = torch.tensor(ref["post__y"], dtype=DT), torch.tensor(ref["post__noise"], dtype=DT)
    mu, var = vae.encode(images)
    assert H.relerr(mu[:, 3], y) < 1e-12 and H.relerr(var[:, 3], noise) < 1e-12           # mnistVAE.encode as the reference wrote it
    pm, B, mu_hat, A_hat = svgp.approximate_posterior_params(test_aux, aux, y, noise)
    for n, v in (("p_m", pm), ("B", B), ("mu_hat", mu_hat), ("A_hat", A_hat)):
        assert H.relerr(v, ref["post__" + n]) < 1e-9, n


def test_sprites_kernel_matrix_and_aux_data_equal_the_reference_code(ref):
    """spritesSVGP.kernel_matrix (SVGPVAE_model.py:489-600; linear x linear, cosine-normalised, SE x SE; inducing x inducing, batch x
    inducing with the GPLVM gather, diagonal) and aux_data_SVGPVAE_sprites (:1086-1115) as the reference wrote them."""
    from oracle import sprites_oracle as SO
    ip, table = torch.tensor(ref["sp__ip"], dtype=DT), torch.tensor(ref["sp__table"], dtype=DT)
    cv, ids = torch.tensor(ref["sp__cv_frames"], dtype=DT), torch.tensor(ref["sp__ids"], dtype=DT)
    b, frames = cv.shape[0], 4
    seg, rep = SO.aux_data_sprites_utils(b, frames, frames)
    means = torch.stack([cv[torch.as_tensor(seg) == g].mean(0) for g in range(b // frames)])
    aux = torch.cat([ids[:, None], torch.repeat_interleave(means, torch.as_tensor(rep), dim=0)], 1)      # what SO.aux_data_SVGPVAE_sprites forms
    assert H.relerr(aux, ref["sp__aux"]) < 1e-14
    se = dict(l_action=torch.tensor(5.0, dtype=DT), sigma_action=torch.tensor(1.4, dtype=DT),
              l_character=torch.tensor(7.0, dtype=DT), sigma_character=torch.tensor(1.2, dtype=DT))
    for tag, kw in (("lin", dict()), ("cos", dict(K_obj_normalize=True)), ("se", dict(K_SE=True, se_params=se))):
        sv = SO.SpritesSVGP(ip, table, 0.01, 100.0, 8, **kw)
        assert H.relerr(sv.kernel_matrix(ip, ip), ref[f"sp__{tag}__K_mm"]) < 1e-13, tag
        assert H.relerr(sv.kernel_matrix(aux, ip, x_inducing=False), ref[f"sp__{tag}__K_nm"]) < 1e-13, tag
        assert H.relerr(sv.kernel_matrix(aux, aux, False, False, diag_only=True), ref[f"sp__{tag}__K_nn_diag"]) < 1e-13, tag
