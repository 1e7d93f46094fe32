"""Deep SVIGP_Hensman baseline (SVIGP_Hensman_model.py) on the HIP library against the literal oracle."""
import numpy as np
import pytest
import torch

from oracle import svigp_oracle as SV
from oracle import svgpvae_oracle as O
from tests import helpers as H
from tests.test_svigp_oracle import svigp_problem

pytestmark = pytest.mark.gpu
DT = torch.float64


def _engine(params, b, N_train, jitter, fixed_ip=False, fixed_gp=False, normalize=False, lr=1e-3):
    from svgp_vae_amd import SVIGP_Hensman_model as SM
    L = params["loc"].shape[0]
    ov = params.get("object_vectors")
    sv = SM.SVIGP_Hensman(fixed_ip, params["inducing_index_points"].numpy(), "main", jitter, N_train, np.float64, L, fixed_gp,
                          None if ov is None else ov.numpy(), normalize)
    return SM.SvigpStepEngine(SM.SVIGP_Hensman_decoder(L), sv, b_max=b, lr=lr, params=params), sv


@pytest.mark.parametrize("case", ["toy", "cfg2_shape", "large_m", "no_table_normalized"])
def test_step_matches_oracle(case):
    kw = dict(b=20, m=7, L=3, M=4, n_obj=9)
    jitter, N, normalize = 1e-6, 500.0, False
    if case == "cfg2_shape":
        kw, N = dict(b=256, m=32, L=16, M=8, n_obj=400), 4050.0
    elif case == "large_m":
        kw, jitter = dict(b=90, m=72, L=2, M=16, n_obj=30), 1e-4
    params, images, aux = svigp_problem(seed=1, **kw)
    if case == "no_table_normalized":
        params.pop("object_vectors"); normalize = True
    L = kw["L"]
    out, grads = SV.loss_and_grads(params, images, aux, jitter=jitter, N_train=N, L=L, K_obj_normalize=normalize)
    eng, _ = _engine(params, kw["b"], N, jitter, normalize=normalize)
    eng.step(images.cuda(), aux.cuda(), adam=False)
    got = eng.outputs()
    names = ("elbo", "recon_loss", "KL_term", "inside_elbo", "recon_images", "inside_recon", "inside_kl", "mean_vectors")
    bad = []
    for i, n in enumerate(names):
        e = H.relerr(got[i], out[i])
        if not e < 1e-8:
            bad.append(f"{n}: {e:.2e}")
    g = eng.grads()
    for k, want in grads.items():
        e = H.relerr(g[k].reshape(-1), want.reshape(-1))
        if not e < 2e-7:
            bad.append(f"grad {k}: {e:.2e}")
    for k in g:
        if k.startswith("enc_"):
            assert float(g[k].abs().max()) == 0.0
    assert not bad, "\n".join(bad)


def test_three_adam_steps_and_prediction_match_oracle():
    params, images, aux = svigp_problem(b=24, m=8, L=3, M=4, n_obj=9, seed=3)
    L, N, jitter = 3, 300.0, 1e-6
    eng, sv = _engine(params, 24, N, jitter, lr=1e-3)
    q = {k: v.clone() for k, v in params.items()}
    keys = None
    ms = vs = None
    for t in range(1, 4):
        out, g = SV.loss_and_grads(q, images, aux, jitter=jitter, N_train=N, L=L)
        if keys is None:
            keys = list(g)
            ms, vs = {k: torch.zeros_like(q[k]) for k in keys}, {k: torch.zeros_like(q[k]) for k in keys}
        O.adam_tf1_step({k: q[k] for k in keys}, g, ms, vs, t, 1e-3)
        eng.step(images.cuda(), aux.cuda(), adam=True)
        assert abs(eng.scalars()["elbo"] - float(out[0])) < 1e-8 * abs(float(out[0]))
    assert eng.scalars()["adam_t"] == 3.0
    p = dict(eng.mn.params); p.update(eng.vp)
    for k in keys:
        assert H.relerr(p[k].reshape(-1), q[k].reshape(-1)) < 1e-8, k
    # conditional generation at new index points (predict_deep_SVIGP_Hensman)
    _, timg, taux = svigp_problem(b=11, m=8, L=3, M=4, n_obj=9, seed=9)
    vae, svgp = SV.make_models(q, jitter, N, L)
    rec, loss = SV.predict_deep_SVIGP_Hensman((timg, taux), vae, svgp)
    from svgp_vae_amd import SVIGP_Hensman_model as SM
    grec, gloss = SM.predict_deep_SVIGP_Hensman((timg.cuda(), taux.cuda()), SM.SVIGP_Hensman_decoder(L), sv)
    assert H.relerr(grec, rec) < 1e-8 and abs(float(gloss) - float(loss)) < 1e-8 * abs(float(loss))


def test_fixed_groups_get_no_update():
    params, images, aux = svigp_problem(seed=5)
    eng, _ = _engine(params, 20, 500.0, 1e-6, fixed_ip=True, fixed_gp=True)
    eng.step(images.cuda(), aux.cuda(), adam=True)
    eng.stream.synchronize()
    p = eng.mn.params
    assert torch.equal(p["inducing_index_points"].cpu(), params["inducing_index_points"])
    assert float(p["l_GP"]) == float(params["l_GP"]) and float(p["amplitude"]) == float(params["amplitude"])
    assert not torch.equal(eng.vp["loc"].cpu(), params["loc"])


def test_cli_driver_trains_and_reports_cgen(golden, tmp_path):
    """`MNIST_experiment.py --elbo SVIGP_Hensman ...` counterpart end to end on a small split of the reference's eval
    images: a few epochs lower the train MSE; the conditional-generation metric is finite and logged."""
    import glob
    import pickle
    from svgp_vae_amd import MNIST_experiment as E
    gin, _ = golden
    d = str(tmp_path) + "/"
    for name, sl in (("train_data3.p", slice(0, 512)), ("eval_data3.p", slice(512, 576)), ("test_data3.p", slice(576, 640))):
        pickle.dump({"images": gin["images"][sl], "aux_data": gin["aux"][sl]}, open(d + name, "wb"))
    pickle.dump(gin["object_vectors"], open(d + "pca_ov_init3.p", "wb"))
    log = E.main(["--elbo", "SVIGP_Hensman", "--mnist_data_path", d, "--train_file", d + "train_data3.p", "--ip_joint",
                  "--GP_joint", "--ov_joint", "--PCA", "--opt_regime", "joint-6", "--eval_every", "3", "--save", "--base_dir", d,
                  "--lr", "0.003"])
    assert len(log["elbo"]) == 6 and all(np.isfinite(log["elbo"]))
    assert log["recon_loss"][-1] < log["recon_loss"][0] and log["elbo"][-1] > log["elbo"][0]
    assert len(log["cgen_mse"]) == 2 and all(np.isfinite(v) and v > 0 for _, v in log["cgen_mse"])
    files = glob.glob(d + "debug_MNIST/*/pics/test_metrics.txt")
    assert files and len(open(files[0]).read().strip().splitlines()) == 2
