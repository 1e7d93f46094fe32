"""The reference-named Python surface (SVGPVAE_model.py / VAE_utils.py names) on the GPU, against
the oracle's restatement of the same functions.  Reads like a test of the reference would."""
import math

import numpy as np
import pytest
import torch

from oracle import svgpvae_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
DT = torch.float64


def _models(golden):
    from svgp_vae_amd.SVGPVAE_model import mnistSVGP
    from svgp_vae_amd.VAE_utils import mnistVAE
    params, images, aux, eps = H.golden_problem(golden)
    VAE = mnistVAE(L=16)
    VAE.params = {k: params[k].clone() for k in VAE.params}
    SVGP_ = mnistSVGP(titsias=False, fixed_inducing_points=False,
                      initial_inducing_points=params["inducing_index_points"].numpy(), fixed_gp_params=False,
                      object_vectors_init=params["object_vectors"].numpy(), name='main', jitter=1e-6,
                      N_train=4050, L=16, K_obj_normalize=False)
    return params, images, aux, eps, VAE, SVGP_


def _oracle_models(params):
    return O.make_models(params, False, 1e-6, 4050.0, 16)


def test_mnistVAE_encode_decode(golden):
    params, images, aux, eps, VAE, _ = _models(golden)
    ovae, _ = _oracle_models(params)
    mu, var = VAE.encode(images[:64])
    omu, ovar = ovae.encode(images[:64])
    assert H.relerr(mu, omu) < 1e-12 and H.relerr(var, ovar) < 1e-12
    rec = VAE.decode(omu)
    assert rec.shape == (64, 28, 28, 1)
    assert H.relerr(rec, ovae.decode(omu)) < 1e-12


def test_mnistSVGP_kernel_matrix_and_channel_methods(golden):
    params, images, aux, eps, VAE, SVGP_ = _models(golden)
    _, osv = _oracle_models(params)
    ip = SVGP_.inducing_index_points
    x = aux[:100]
    assert H.relerr(SVGP_.kernel_matrix(ip, ip), osv.kernel_matrix(params["inducing_index_points"],
                                                                    params["inducing_index_points"])) < 1e-13
    assert H.relerr(SVGP_.kernel_matrix(x, ip, x_inducing=False),
                    osv.kernel_matrix(x, params["inducing_index_points"], x_inducing=False)) < 1e-13
    assert H.relerr(SVGP_.kernel_matrix(x, x, x_inducing=False, y_inducing=False, diag_only=True),
                    osv.kernel_matrix(x, x, False, False, diag_only=True)) < 1e-13
    with pytest.raises(NotImplementedError):
        SVGP_.kernel_matrix(x, x, x_inducing=False, y_inducing=False)
    y = torch.randn(100, dtype=DT, generator=torch.Generator().manual_seed(0))
    noise = torch.rand(100, dtype=DT, generator=torch.Generator().manual_seed(1)) + 0.05
    mean, B, mu_hat, A_hat = SVGP_.approximate_posterior_params(x, x, y, noise)
    omean, oB, omu_hat, oA_hat = osv.approximate_posterior_params(x, x, y, noise)
    for a, b in ((mean, omean), (B, oB), (mu_hat, omu_hat), (A_hat, oA_hat)):
        assert H.relerr(a, b) < 1e-9
    l3, kl = SVGP_.variational_loss(x, y, mu_hat, A_hat, noise)
    ol3, okl = osv.variational_loss(x, y, omu_hat, oA_hat, noise)
    assert abs(float(l3) - float(ol3)) < 1e-9 * abs(float(ol3))
    assert abs(float(kl) - float(okl)) < 1e-9 * abs(float(okl))
    assert len(SVGP_.variable_summary()) == 4


@pytest.mark.parametrize("GECO", [False, True])
def test_forward_pass_SVGPVAE_sixteen_tuple(golden, GECO):
    from svgp_vae_amd.SVGPVAE_model import forward_pass_SVGPVAE, gradients_SVGPVAE
    params, images, aux, eps, VAE, SVGP_ = _models(golden)
    ovae, osv = _oracle_models(params)
    kw = dict(beta=0.001, C_ma=0.01, lagrange_mult=1.3, alpha=0.99, kappa=math.sqrt(0.020), clipping_qs=True,
              GECO=GECO)
    got = forward_pass_SVGPVAE((images, aux), vae=VAE, svgp=SVGP_, epsilon=eps, **kw)
    want = O.forward_pass_SVGPVAE((images, aux), kw["beta"], ovae, osv, torch.tensor(0.01, dtype=DT),
                                  torch.tensor(1.3, dtype=DT), 0.99, kw["kappa"], clipping_qs=True, GECO=GECO,
                                  epsilon=eps, formulation="literal")
    assert len(got) == 16
    for i, (a, b) in enumerate(zip(got, want)):
        assert H.relerr(a, b) < 1e-8, f"tuple member {i}"
    g = gradients_SVGPVAE(VAE, SVGP_)
    _, og = O.loss_and_grads(params, images, aux, eps, beta=0.001, C_ma=torch.tensor(0.01, dtype=DT),
                             lagrange_mult=torch.tensor(1.3, dtype=DT), alpha=0.99, kappa=kw["kappa"],
                             clipping_qs=True, GECO=GECO, jitter=1e-6, N_train=4050.0, L=16, formulation="efficient")
    for k in og:
        assert H.relerr(g[k], og[k]) < 1e-7, k


def test_train_step_SVGPVAE_updates_the_models_in_place(golden):
    """Three optimiser steps through the reference-named surface reproduce the trajectory fixture
    (GECO first-step alpha=0, state carry, TF1 Adam, ragged last batch)."""
    from svgp_vae_amd.SVGPVAE_model import train_step_SVGPVAE
    gin, gout = golden
    params, _, _, _, VAE, SVGP_ = _models(golden)
    for t, r in enumerate([slice(0, 256), slice(256, 512), slice(512, 640)]):
        _, images, aux, eps = H.golden_problem(golden, r)
        out = train_step_SVGPVAE((images, aux), 0.001, VAE, SVGP_, alpha=0.99, kappa=math.sqrt(0.020), lr=1e-3,
                                 clipping_qs=True, GECO=True, epsilon=eps)
        assert abs(float(out[0]) - float(gout["geco_traj_elbo"][t])) <= 1e-7 * abs(float(gout["geco_traj_elbo"][t]))
    assert H.relerr(SVGP_.inducing_index_points, gout["geco_traj_param_inducing_index_points"]) < 1e-7
    assert H.relerr(SVGP_.object_vectors, gout["geco_traj_param_object_vectors"]) < 1e-7
    assert H.relerr(VAE.params["dec_c2_w"], gout["geco_traj_param_dec_c2_w"]) < 1e-7
    assert float(SVGP_.l_GP) != 1.0 and float(SVGP_.amplitude) != 1.0


def test_not_yet_built_branches_say_so():
    from svgp_vae_amd.MNIST_experiment import main
    with pytest.raises(NotImplementedError, match="CVAE"):
        main(["--elbo", "CVAE"])


def test_bacthing_predict_conditional_generation(golden):
    """Test-MSE path (SVGPVAE_model.py:1026-1083, MNIST_experiment.py:457-486): encode a 'train' set in
    batches, predict held-out rows from the GP posterior, compare with the oracle's literal restatement."""
    from svgp_vae_amd.SVGPVAE_model import bacthing_predict_SVGPVAE_rotated_mnist, batching_encode_SVGPVAE
    gin, _ = golden
    params, _, _, _, VAE, SVGP_ = _models(golden)
    SVGP_.N_train = 512.0
    ovae, osv = O.make_models(params, False, 1e-6, 512.0, 16)
    img = torch.tensor(gin["images"], dtype=DT)
    aux = torch.tensor(gin["aux"], dtype=DT)
    eps = torch.tensor(gin["epsilon"][512:640], dtype=DT)
    means, vars_ = [], []
    for lo in (0, 256):                                   # batches of 256 like the reference's loop
        mu, var, _ = batching_encode_SVGPVAE((img[lo:lo + 256], aux[lo:lo + 256]), VAE, clipping_qs=True)
        means.append(mu); vars_.append(var)
    means, vars_ = torch.cat(means), torch.cat(vars_)
    omu, ovar, _ = O.batching_encode_SVGPVAE((img[:512], aux[:512]), ovae, clipping_qs=True)
    assert H.relerr(means, omu) < 1e-12 and H.relerr(vars_, ovar) < 1e-12
    recon, loss = bacthing_predict_SVGPVAE_rotated_mnist((img[512:640], aux[512:640]), VAE, SVGP_, means, vars_,
                                                         aux[:512], epsilon=eps)
    orecon, oloss = O.bacthing_predict_SVGPVAE_rotated_mnist((img[512:640], aux[512:640]), ovae, osv, omu, ovar,
                                                             aux[:512], epsilon=eps)
    assert recon.shape == (128, 28, 28, 1)
    assert H.relerr(recon, orecon) < 1e-8
    assert abs(float(loss) - float(oloss)) < 1e-8 * float(oloss)
    # MSE_cgen as the driver computes it: sum of batch losses / N_test
    assert float(loss) / 128 > 0


@pytest.mark.parametrize("elbo", ["SVGPVAE_Hensman", "SVGPVAE_Titsias"])
def test_cli_driver_trains_and_reports_cgen(golden, tmp_path, elbo):
    """`MNIST_experiment.py --elbo SVGPVAE_Hensman | SVGPVAE_Titsias ...` counterpart end to end on a small split of the
    reference's eval images: 6 epochs of GECO training must lower the train MSE, and the eval / conditional
    generation metrics must be finite and written to pics/test_metrics.txt."""
    import pickle
    from svgp_vae_amd import MNIST_experiment as E
    gin, _ = golden
    d = str(tmp_path) + "/"
    for name, sl in (("train_data3.p", slice(0, 512)), ("eval_data3.p", slice(512, 576)), ("test_data3.p", slice(576, 640))):
        pickle.dump({"images": gin["images"][sl], "aux_data": gin["aux"][sl]}, open(d + name, "wb"))
    pickle.dump(gin["object_vectors"], open(d + "pca_ov_init3.p", "wb"))
    args = E.build_parser().parse_args(
        ["--elbo", elbo, "--mnist_data_path", d, "--train_file", d + "train_data3.p", "--ip_joint",
         "--GP_joint", "--ov_joint", "--clip_qs", "--GECO", "--PCA", "--opt_regime", "joint-6", "--eval_every", "3",
         "--save", "--base_dir", d, "--lr", "0.003"])
    log = E.run_experiment_rotated_mnist_SVGPVAE(args)
    assert len(log["elbo"]) == 6 and all(np.isfinite(log["elbo"]))
    assert log["recon_loss"][-1] < log["recon_loss"][0]
    assert len(log["cgen_mse"]) == 2 and all(np.isfinite(v) and v > 0 for _, v in log["cgen_mse"])
    import glob
    files = glob.glob(d + "debug_MNIST/*/pics/test_metrics.txt")
    assert files and len(open(files[0]).read().strip().splitlines()) == 2
    with pytest.raises(NotImplementedError):
        E.main(["--elbo", "VAE"])
