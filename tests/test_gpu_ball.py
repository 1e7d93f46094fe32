"""Moving-ball SVGP-VAE on the HIP library against the literal CPU restatement (oracle/ball_oracle.py):
BALL_experiment.py --elbo SVGPVAE_Hensman | SVGPVAE_Titsias (SVGPVAE_model.py:17-171, 638-715)."""
import numpy as np
import pytest
import torch

from oracle import ball_oracle as BO
from oracle import pearce_vae_oracle as PO
from tests import helpers as H

pytestmark = pytest.mark.gpu
DT = torch.float64


def _problem(batch, T, px, hidden, m, seed=0, lt=2.0):
    g = torch.Generator().manual_seed(seed)
    vid = PO.make_video_batch(tmax=T, px=px, py=px, lt=lt, batch=batch, r=max(2, px // 10), generator=g, dtype=DT)
    p = {k: v.to(DT) for k, v in PO.init_mlp_params(px, px, hidden=hidden, seed=seed).items()}
    # non-zero biases so that their gradients / updates are exercised from a generic point
    for k in ("encB1", "encB2", "decB1", "decB2"):
        p[k] = 0.05 * torch.randn(*p[k].shape, dtype=DT, generator=g)
    for c in "xy":
        p[f"ip_{c}"] = BO.BallSVGP.initial_inducing_points(m, False, 1, T, 1, T) + 0.1 * torch.randn(m, dtype=DT, generator=g)
        p[f"l_{c}"] = torch.tensor(lt + (0.3 if c == "y" else 0.0), dtype=DT)
    eps = torch.randn(batch, T, 2, dtype=DT, generator=g)
    return p, vid, eps


def _engine(p, batch, T, px, hidden, m, *, titsias, jitter, clip_qs, beta, fixed_ip=False, fixed_gp=False, **kw):
    from svgp_vae_amd import ball
    mk = lambda n: ball.SVGP(titsias, m, fixed_ip, 1, T, 2.0, fixed_gp, n, jitter, 1, T, 2.0)
    flat = {k: (v.reshape(-1) if k.startswith(("encB", "decB", "l_")) else v) for k, v in p.items()}
    return ball.BallStepEngine(mk("x"), mk("y"), batch=batch, tmax=T, px=px, py=px, hidden=hidden, clip_qs=clip_qs,
                               beta=beta, params=flat, **kw)


CASES = {
    "small_hensman": dict(shape=(5, 12, 8, 16, 6), titsias=False, jitter=1e-6, clip_qs=False),
    "small_hensman_clip": dict(shape=(4, 10, 8, 12, 5), titsias=False, jitter=1e-6, clip_qs=True),
    "small_titsias": dict(shape=(5, 12, 8, 16, 6), titsias=True, jitter=1e-6, clip_qs=True),
    "reference_shape_hensman": dict(shape=(35, 30, 32, 500, 15), titsias=False, jitter=1e-6, clip_qs=True),
    "reference_shape_titsias": dict(shape=(35, 30, 32, 500, 15), titsias=True, jitter=1e-6, clip_qs=True),
}


@pytest.mark.parametrize("case", list(CASES))
def test_step_matches_oracle(case):
    cs = CASES[case]
    batch, T, px, hidden, m = cs["shape"]
    p, vid, eps = _problem(batch, T, px, hidden, m, seed=3)
    beta = 0.8
    out, loss, grads = BO.loss_and_grads(p, vid, eps, beta=beta, titsias=cs["titsias"], jitter=cs["jitter"],
                                         clipping_qs=cs["clip_qs"])
    eng = _engine(p, batch, T, px, hidden, m, titsias=cs["titsias"], jitter=cs["jitter"], clip_qs=cs["clip_qs"], beta=beta)
    eng.step(vid.cuda(), eps.cuda(), adam=False)
    got = eng.outputs()
    names = ("elbo", "recon", "KL_term", "inside_elbo", "ce_term", "full_p_mu", "full_p_var", "qnet_mu", "qnet_var",
             "pred_vid", "l_x", "l_y", "inside_recon", "inside_kl", "ip_x", "ip_y", "cov_mean_x", "cov_mean_y")
    bad = []
    for i, n in enumerate(names):
        want = out[i] if torch.is_tensor(out[i]) else torch.tensor(float(out[i]), dtype=DT)
        if n == "inside_kl" and cs["titsias"]:
            want = torch.zeros(batch, dtype=DT)
        e = H.relerr(got[i], want)
        if not e < 1e-8:
            bad.append(f"{n}: {e:.2e}")
    sc = eng.scalars()
    if not abs(sc["elbo"] - float(out[0].mean())) <= 1e-9 * abs(float(out[0].mean())):
        bad.append(f"mean elbo {sc['elbo']} vs {float(out[0].mean())}")
    eng.stream.synchronize()
    for k in BO.PARAM_ORDER:
        e = H.relerr(eng.grads[k].reshape(-1), grads[k].reshape(-1))
        if not e < 1e-7:
            bad.append(f"grad {k}: {e:.2e}")
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("titsias", [False, True])
def test_three_adam_steps_match_oracle_trajectory(titsias):
    batch, T, px, hidden, m = 6, 10, 8, 16, 5
    p, _, _ = _problem(batch, T, px, hidden, m, seed=5)
    g = torch.Generator().manual_seed(11)
    vids = [PO.make_video_batch(tmax=T, px=px, py=px, lt=2.0, batch=batch, r=2, generator=g, dtype=DT) for _ in range(3)]
    epss = [torch.randn(batch, T, 2, dtype=DT, generator=g) for _ in range(3)]
    want, elbos = BO.train_trajectory(p, vids, epss, beta=1.0, titsias=titsias, jitter=1e-6, clipping_qs=True, lr=1e-3,
                                      clip_grad=True, train_ip=True, train_gp=False)
    eng = _engine(p, batch, T, px, hidden, m, titsias=titsias, jitter=1e-6, clip_qs=True, beta=1.0, fixed_gp=True,
                  clip_grad=True, lr=1e-3)
    got_elbo = []
    for v, e in zip(vids, epss):
        eng.step(v.cuda(), e.cuda(), adam=True)
        got_elbo.append(eng.scalars()["elbo"])
    assert np.allclose(got_elbo, elbos, rtol=1e-8)
    assert eng.scalars()["adam_t"] == 3.0
    for k in BO.PARAM_ORDER:
        assert H.relerr(eng.params[k].reshape(-1), want[k].reshape(-1)) < 1e-8, k
    # fixed GP parameters stay put (BALL_experiment.py:103: constants when not --GP_joint)
    assert float(eng.params["l_x"][0]) == float(p["l_x"]) and float(eng.params["l_y"][0]) == float(p["l_y"])


def test_on_device_samples_differ_between_coordinates_and_steps():
    batch, T, px, hidden, m = 8, 16, 8, 16, 6
    p, vid, _ = _problem(batch, T, px, hidden, m, seed=7)
    eng = _engine(p, batch, T, px, hidden, m, titsias=False, jitter=1e-6, clip_qs=True, beta=1.0)
    eng.step(vid.cuda(), None, adam=False)
    eng.stream.synchronize()
    ex, ey = eng._v(0, "eps", (T, batch)).clone(), eng._v(1, "eps", (T, batch)).clone()
    assert float((ex - ey).abs().max()) > 0.1
    eng.step(vid.cuda(), None, adam=False)
    eng.stream.synchronize()
    assert float((eng._v(0, "eps", (T, batch)) - ex).abs().max()) > 0.1
    allv = torch.cat([ex.reshape(-1), ey.reshape(-1)])
    assert abs(float(allv.mean())) < 0.3 and 0.7 < float(allv.std()) < 1.3


def test_elbo_graph_builder_surface_and_device_video_source():
    from svgp_vae_amd import ball
    src = ball.VideoBatchSource(tmax=12, px=16, py=16, lt=2, batch=4, seed=1, r=3)
    vid = src()
    torch.cuda.synchronize()
    assert vid.shape == (4, 12, 16, 16) and set(np.unique(vid.cpu().numpy())) <= {0.0, 1.0}
    area = vid.sum((2, 3))
    assert 20 <= float(area.max()) <= 32                              # a radius-3 disc covers about 28 pixels
    vid2 = src()
    assert float((vid2 - vid).abs().sum()) > 0
    mk = lambda n: ball.SVGP(False, 5, False, 1, 12, 2.0, False, n, 1e-6, 1, 12, 2.0)
    out = ball.build_SVGPVAE_elbo_graph(vid, 1.0, mk("x"), mk("y"), clipping_qs=True)
    assert len(out) == 19 and out[0].shape == (4,) and out[5].shape == (4, 12, 2) and out[9].shape == (4, 12, 16, 16)
    assert torch.isfinite(out[0]).all()
    assert torch.allclose(out[0], out[1] + 1.0 * out[2])
    assert torch.allclose(out[2], out[4] + out[3])


def test_unsupported_shapes_fail_loudly():
    from svgp_vae_amd import _lib, ball
    mk = lambda n, m: ball.SVGP(False, m, False, 1, 30, 2.0, False, n, 1e-6, 1, 30, 2.0)
    with pytest.raises(_lib.SvgpError):
        ball.BallStepEngine(mk("x", 80), mk("y", 80), batch=4, tmax=30, px=8, py=8, hidden=8)     # m > 64
    with pytest.raises(_lib.SvgpError):
        ball.BallStepEngine(mk("x", 8), mk("y", 8), batch=70, tmax=30, px=8, py=8, hidden=8)      # > 64 videos


# ---------------------------------------------------------------------------------------------------------
# Pearce baseline (BASELINE configs[0]: BALL_experiment.py --elbo VAE; also GPVAE_Pearce and NP)
# ---------------------------------------------------------------------------------------------------------
def _pearce_engine(p, type_elbo, lt, GP_joint, batch, T, px, hidden, beta, **kw):
    from svgp_vae_amd import ball
    flat = {k: (v.reshape(-1) if k.startswith(("encB", "decB", "l_")) else v) for k, v in p.items()}
    return ball.PearceStepEngine(type_elbo, lt, 0.5, GP_joint, 2.0, batch=batch, tmax=T, px=px, py=px, hidden=hidden,
                                 beta=beta, params=flat, **kw)


PEARCE_CASES = {
    "vae_small": dict(shape=(5, 12, 8, 16), type_elbo="VAE", lt=0.001, joint=False),
    "pearce_small_joint": dict(shape=(5, 12, 8, 16), type_elbo="GPVAE_Pearce", lt=2.0, joint=True),
    "np_small_joint": dict(shape=(6, 14, 8, 16), type_elbo="NP", lt=2.0, joint=True),
    "vae_config1_shape": dict(shape=(35, 30, 32, 500), type_elbo="VAE", lt=0.001, joint=False),
    "pearce_config1_shape": dict(shape=(35, 30, 32, 500), type_elbo="GPVAE_Pearce", lt=2.0, joint=False),
    "np_config1_shape": dict(shape=(35, 30, 32, 500), type_elbo="NP", lt=2.0, joint=False),
}


@pytest.mark.parametrize("case", list(PEARCE_CASES))
def test_pearce_step_matches_oracle(case):
    cs = PEARCE_CASES[case]
    batch, T, px, hidden = cs["shape"]
    p, vid, eps = _problem(batch, T, px, hidden, 4, seed=4)
    p = {k: v for k, v in p.items() if not k.startswith("ip_")}
    lt = cs["lt"]
    p["l_x"] = torch.tensor(lt * (1.2 if cs["joint"] else 1.0), dtype=DT)
    p["l_y"] = torch.tensor(lt * (0.9 if cs["joint"] else 1.0), dtype=DT)
    ran_ind = con_tf = None
    if cs["type_elbo"] == "NP":
        g = torch.Generator().manual_seed(9)
        ran_ind = torch.stack([torch.randperm(T, generator=g) for _ in range(batch)])
        con_tf = T // 2 - 1
    beta = 0.9
    out, loss, grads = BO.pearce_loss_and_grads(p, vid, eps, beta=beta, type_elbo=cs["type_elbo"], lt=lt, ran_ind=ran_ind,
                                                con_tf=con_tf)
    eng = _pearce_engine(p, cs["type_elbo"], lt, cs["joint"], batch, T, px, hidden, beta)
    eng.step(vid.cuda(), eps.cuda(), adam=False, ran_ind=None if ran_ind is None else ran_ind.numpy(), con_tf=con_tf)
    got = eng.outputs()
    names = ("elbo", "recon", "prior_kl", "full_p_mu", "full_p_var", "qnet_mu", "qnet_var", "pred_vid")
    bad = []
    for i, n in enumerate(names):
        e = H.relerr(got[i], out[i])
        if not e < 1e-8:
            bad.append(f"{n}: {e:.2e}")
    if not abs(eng.scalars()["elbo"] - float(out[0].mean())) <= 1e-9 * abs(float(out[0].mean())):
        bad.append("mean elbo")
    eng.stream.synchronize()
    for k in BO.PEARCE_PARAM_ORDER:
        if k.startswith("l_") and not cs["joint"]:
            assert float(eng.grads[k].abs().max()) == 0.0          # constants when not --GP_joint
            continue
        e = H.relerr(eng.grads[k].reshape(-1), grads[k].reshape(-1))
        if not e < 1e-7:
            bad.append(f"grad {k}: {e:.2e}")
    assert not bad, "\n".join(bad)


def test_vae_limit_is_the_closed_form_standard_vae():
    """lt = 0.001 (BALL_experiment.py:46-48): prior KL term = -KL(N(p_m, p_v) || N(0, 1)) summed over frames and
    coordinates, with p_m = y/(1+var), p_v = var/(1+var)."""
    batch, T, px, hidden = 6, 10, 8, 16
    p, vid, eps = _problem(batch, T, px, hidden, 4, seed=6)
    p = {k: v for k, v in p.items() if not k.startswith("ip_")}
    p["l_x"] = p["l_y"] = torch.tensor(0.001, dtype=DT)
    eng = _pearce_engine(p, "VAE", 0.001, False, batch, T, px, hidden, 1.0)
    eng.step(vid.cuda(), eps.cuda(), adam=False, backward=False)
    elbo, recon, kl, p_m, p_v, q_m, q_v = [t.cpu() for t in eng.outputs()[:7]]
    assert torch.allclose(p_m, q_m / (1 + q_v), atol=1e-12) and torch.allclose(p_v, q_v / (1 + q_v), atol=1e-12)
    want = -torch.distributions.kl_divergence(torch.distributions.Normal(p_m, p_v.sqrt()),
                                              torch.distributions.Normal(torch.zeros_like(p_m), torch.ones_like(p_m))).sum((1, 2))
    assert torch.allclose(kl, want, rtol=1e-9, atol=1e-9)
    assert torch.allclose(elbo, recon + kl)


def test_pearce_three_adam_steps_raise_the_elbo_and_track_the_oracle():
    batch, T, px, hidden = 6, 10, 8, 16
    p, vid, eps = _problem(batch, T, px, hidden, 4, seed=8)
    p = {k: v for k, v in p.items() if not k.startswith("ip_")}
    p["l_x"], p["l_y"] = torch.tensor(2.0, dtype=DT), torch.tensor(2.0, dtype=DT)
    eng = _pearce_engine(p, "GPVAE_Pearce", 2.0, True, batch, T, px, hidden, 1.0, lr=1e-3)
    from oracle import svgpvae_oracle as O
    q = {k: v.clone() for k, v in p.items()}
    ms, vs = {k: torch.zeros_like(v) for k, v in q.items()}, {k: torch.zeros_like(v) for k, v in q.items()}
    for t in range(1, 4):
        out, loss, g = BO.pearce_loss_and_grads(q, vid, eps, beta=1.0, type_elbo="GPVAE_Pearce", lt=2.0)
        O.adam_tf1_step(q, g, ms, vs, t, 1e-3)
        eng.step(vid.cuda(), eps.cuda(), adam=True)
        assert abs(eng.scalars()["elbo"] - float(out[0].mean())) < 1e-8 * abs(float(out[0].mean()))
    for k in BO.PEARCE_PARAM_ORDER:
        assert H.relerr(eng.params[k].reshape(-1), q[k].reshape(-1)) < 1e-8, k


@pytest.mark.parametrize("elbo,extra", [("VAE", []), ("GPVAE_Pearce", ["--GP_joint"]), ("NP", []),
                                         ("SVGPVAE_Hensman", ["--clip_qs", "--GP_joint", "--ip_joint", "--jitter", "1e-6"]),
                                         ("SVGPVAE_Titsias", ["--clip_qs", "--jitter", "1e-6"])])
def test_ball_cli_end_to_end(tmp_path, elbo, extra):
    """BALL_experiment.py driver: reference flags, fresh synthetic batch per step, test-batch diagnostics, log file."""
    from svgp_vae_amd import BALL_experiment as BE
    argv = ["--elbo", elbo, "--steps", "8", "--eval_every", "4", "--hidden", "32", "--tmax", "12", "--m", "6", "--ip_max",
            "12", "--base_dir", str(tmp_path), "--save", "--save_model", "--seed", "3"] + extra
    log = BE.main(argv)
    assert [r["Step"] for r in log] == [4, 8]
    for r in log:
        assert np.isfinite(r["elbo"]) and np.isfinite(r["MSE"]) and r["min q_var"] > 0
    runs = [d for d in tmp_path.iterdir() if d.is_dir()]
    assert len(runs) == 1 and (runs[0] / "res" / "ELBO_log.jsonl").exists() and (runs[0] / "model.pt").exists()
    assert (tmp_path / "Test_Batches_2_12.pkl").exists()


def test_vae_training_improves_the_test_elbo():
    """BASELINE configs[0] (`BALL_experiment.py --elbo VAE`) at full shape: 60 Adam steps on fresh synthetic batches
    raise the ELBO of the fixed test batch by a wide margin (untrained: about -35 * 30 * 1024 * log 2)."""
    from svgp_vae_amd import BALL_experiment as BE
    args = BE.build_parser().parse_args(["--elbo", "VAE", "--steps", "60", "--seed", "0"])
    eng = BE.build_engine(args)
    TT, TD = __import__("svgp_vae_amd.ball", fromlist=["x"]).Make_Video_batch(tmax=30, px=32, py=32, lt=2, batch=35, seed=0, r=3)
    before = BE.evaluate(eng, TT, TD, 1.0)["elbo"]
    src = __import__("svgp_vae_amd.ball", fromlist=["x"]).VideoBatchSource(tmax=30, px=32, py=32, lt=2, batch=35, seed=1, r=3)
    for _ in range(60):
        v = src(); v.record_stream(eng.stream)
        eng.step(v, None, adam=True)
    after = BE.evaluate(eng, TT, TD, 1.0)["elbo"]
    assert eng.scalars()["adam_t"] == 60.0
    assert after > before + 0.5 * abs(before), (before, after)
