"""SVIGP_Hensman baseline: the literal restatement equals the efficient form the HIP path implements."""
import torch

from oracle import staged_gp as SG
from oracle import svigp_oracle as SV
from oracle import svgpvae_oracle as O
from tests import helpers as H

DT = torch.float64


def svigp_problem(b=20, m=7, L=3, M=4, n_obj=9, seed=0):
    params, images, aux, eps = H.toy_problem(b=b, m=m, L=L, M=M, n_obj=n_obj, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    params = dict(params)
    params["loc"] = 0.3 * torch.randn(L, m, dtype=DT, generator=g)
    params["scale"] = torch.eye(m, dtype=DT)[None].repeat(L, 1, 1) + 0.1 * torch.randn(L, m, m, dtype=DT, generator=g)
    params["noise"] = torch.tensor(0.35, dtype=DT)
    return params, images, aux


def test_literal_equals_efficient():
    params, images, aux = svigp_problem()
    L = params["loc"].shape[0]
    vae, svgp = SV.make_models(params, 1e-6, 500.0, L)
    out = SV.forward_pass_deep_SVIGP_Hensman((images, aux), vae, svgp)
    K, Kn, knn = SG.kernel_matrix_fwd(aux, params["inducing_index_points"], params.get("object_vectors"), params["l_GP"],
                                      params["amplitude"])
    Z, L3, KL = SV.efficient_terms(K, Kn, knn, params["loc"], params["scale"], params["noise"], 1e-6)
    assert torch.allclose(out[7], Z, rtol=1e-10, atol=1e-12)
    assert abs(float(out[5] - L3)) < 1e-10 * abs(float(L3)) and abs(float(out[6] - KL)) < 1e-10 * abs(float(KL))
    b = images.shape[0]
    assert abs(float(out[3] - (L3 - b / 500.0 * KL))) < 1e-10 * abs(float(out[3]))


def test_prediction_at_train_points_equals_training_mean_vectors_and_gradients_are_finite():
    params, images, aux = svigp_problem(seed=2)
    L = params["loc"].shape[0]
    vae, svgp = SV.make_models(params, 1e-6, 500.0, L)
    out = SV.forward_pass_deep_SVIGP_Hensman((images, aux), vae, svgp)
    rec, loss = SV.predict_deep_SVIGP_Hensman((images, aux), vae, svgp)
    assert torch.allclose(rec, out[4], atol=1e-12)
    assert abs(float(loss) - float(out[1])) < 1e-12 * abs(float(loss))
    _, grads = SV.loss_and_grads(params, images, aux, jitter=1e-6, N_train=500.0, L=L)
    for k, v in grads.items():
        assert torch.isfinite(v).all() and float(v.abs().max()) > 0, k
