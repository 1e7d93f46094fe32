"""Round 6: the decoder's reverse pass as data half (svgp_mnist_decoder_bwd_data) + weight half (svgp_mnist_decoder_bwd_weights;
in the training step: rider workgroups of the reverse factor launch, svgp_gp_factor_bwd_nofinal_wgrad) against the one-kernel
form svgp_mnist_decoder_bwd and against the oracle (tf.gradients of VAE_utils.py:128-141,154-162; MNIST_experiment.py:202-205)."""
import ctypes as C
import math
import os

import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _engine(b, m, L, seed=3, geco=True):
    params, images, aux, eps = H.toy_problem(b=b, m=m, L=L, M=4, n_obj=20, seed=seed)
    eng = H.engine_for(params, b, geco=geco, N_train=400.0)
    dev = eng.device
    eng.bind(images.to(dev), aux.to(dev), eps.to(dev))
    return eng, (params, images, aux, eps)


@pytest.mark.parametrize("b,m,L", [(48, 16, 4), (300, 32, 16), (7, 12, 3)])
def test_two_halves_equal_the_one_kernel_form(b, m, L):
    from svgp_vae_amd import _lib
    eng, _ = _engine(b, m, L)
    eng.run(adam=False)
    eng.synchronize()
    cfg, th, ws, st = C.byref(eng.cfg), eng.theta.data_ptr(), eng.ws.data_ptr(), eng.state.data_ptr()
    img, s = eng._bound[0].data_ptr(), eng.stream.cuda_stream
    n_dec = eng.pl.n_vae - eng.pl.n_enc
    part = lambda: eng.ws_view("part_dec", (eng.wl.n_part, n_dec))
    _lib.call("svgp_mnist_decoder_bwd", cfg, th, img, ws, st, s)
    eng.synchronize()
    p0, z0 = part().clone(), eng.ws_view("zbar", (b, L)).clone()
    eng.ws_view("zbar", (b, L)).zero_()
    _lib.call("svgp_mnist_decoder_bwd_data", cfg, th, img, ws, st, s)
    eng.synchronize()
    assert torch.equal(eng.ws_view("zbar", (b, L)), z0)          # same helper, same order: bit-equal
    scale = float(p0.abs().max())
    for threads, n_types in ((256, 1), (256, 2), (256, 3), (512, 1), (512, 3)):
        part().fill_(float("nan"))
        _lib.call("svgp_mnist_decoder_bwd_weights", cfg, img, ws, st, threads, n_types, s)
        eng.synchronize()
        assert float((part() - p0).abs().max()) < 1e-13 * scale, (threads, n_types)
    part().fill_(float("nan"))
    _lib.call("svgp_gp_factor_bwd_nofinal_wgrad", cfg, img, ws, st, s)       # the riders
    eng.synchronize()
    assert float((part() - p0).abs().max()) < 1e-13 * scale


def test_rider_launch_leaves_the_factor_stage_unchanged():
    from svgp_vae_amd import _lib
    eng, _ = _engine(64, 32, 16)
    eng.run(adam=False)
    eng.synchronize()
    cfg, ws, st = C.byref(eng.cfg), eng.ws.data_ptr(), eng.state.data_ptr()
    img, s = eng._bound[0].data_ptr(), eng.stream.cuda_stream
    shapes = dict(fb_part=(2, 16, 32, 32), vbar=(16, 32), Ssym=(16, 32, 32), Qm=(16, 32, 32))
    _lib.call("svgp_gp_factor_bwd_nofinal", cfg, ws, st, s)
    eng.synchronize()
    ref = {k: eng.ws_view(k, sh).clone() for k, sh in shapes.items()}
    for k, sh in shapes.items():
        eng.ws_view(k, sh).zero_()
    _lib.call("svgp_gp_factor_bwd_nofinal_wgrad", cfg, img, ws, st, s)
    eng.synchronize()
    for k, sh in shapes.items():
        assert torch.equal(eng.ws_view(k, sh), ref[k]), k


@pytest.mark.parametrize("geco", [True, False])
def test_step_gradients_match_oracle_and_the_one_kernel_step(geco, monkeypatch):
    from oracle import svgpvae_oracle as O
    grads = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("SVGP_DEC_SPLIT", flag)
        eng, (params, images, aux, eps) = _engine(48, 16, 4, geco=geco)
        eng.run(adam=False)
        eng.synchronize()
        grads[flag] = {k: v.clone() for k, v in eng.grads().items()}
        elbo = eng.scalars()["elbo"]
    out, og = O.loss_and_grads(params, images, aux, eps, beta=0.001, C_ma=torch.zeros((), dtype=O.DT),
                               lagrange_mult=torch.ones((), dtype=O.DT), alpha=0.0, kappa=math.sqrt(0.02),
                               clipping_qs=True, GECO=geco, jitter=1e-6, N_train=400.0, L=4, formulation="efficient")
    assert abs(elbo - float(out[0])) < 1e-9 * abs(float(out[0]))
    for k, v in og.items():
        assert H.relerr(grads["1"][k], v) < 1e-7, k
        assert H.relerr(grads["1"][k], grads["0"][k]) < 1e-12, k


def test_split_step_is_bitwise_reproducible():
    eng, _ = _engine(256, 32, 16)
    eng.run(adam=False)
    eng.synchronize()
    g0 = {k: v.clone() for k, v in eng.grads().items()}
    for _ in range(3):
        eng.reset_state()                 # the GECO state advances on the device also without Adam
        eng.run(adam=False)
        eng.synchronize()
        for k, v in eng.grads().items():
            assert torch.equal(v, g0[k]), k


@pytest.mark.parametrize("b,m,L,M", [(256, 32, 16, 8), (48, 16, 4, 4), (300, 24, 5, 4)])
def test_encoder_reverse_and_kernel_matrix_vjp_in_one_launch(b, m, L, M):
    """svgp_mnist_encoder_bwd_km == svgp_kernel_matrix_bwd_partials + svgp_mnist_encoder_bwd, bit for bit (the VJP workgroups run
    on the first 256 threads of 512-thread workgroups whose other waves exit at once)."""
    from svgp_vae_amd import _lib
    params, images, aux, eps = H.toy_problem(b=b, m=m, L=L, M=M, n_obj=20, seed=5)
    eng = H.engine_for(params, b, geco=True, N_train=400.0)
    dev = eng.device
    eng.bind(images.to(dev), aux.to(dev), eps.to(dev))
    eng.run(adam=False)
    eng.synchronize()
    cfg, th, ws = C.byref(eng.cfg), eng.theta.data_ptr(), eng.ws.data_ptr()
    img, ax, s = eng._bound[0].data_ptr(), eng._bound[1].data_ptr(), eng.stream.cuda_stream
    nrb = (b + (256 // m) - 1) // (256 // m)
    views = lambda: dict(part_enc=eng.ws_view("part_enc", (eng.wl.n_part, eng.pl.n_enc)), d_on=eng.ws_view("d_on", (b, M)),
                         part_gp=eng.ws_view("part_gp", (m + nrb, 2)), ip=eng.grads()["inducing_index_points"])
    _lib.call("svgp_kernel_matrix_bwd_partials", cfg, th, ax, ws, s)
    _lib.call("svgp_mnist_encoder_bwd", cfg, th, img, ws, s)
    eng.synchronize()
    ref = {k: v.clone() for k, v in views().items()}
    for v in views().values():
        v.fill_(float("nan"))
    _lib.call("svgp_mnist_encoder_bwd_km", cfg, th, img, ax, ws, s)
    eng.synchronize()
    for k, v in views().items():
        assert torch.equal(v, ref[k]), k


def test_precomputed_effective_weights_forms_equal_the_plain_ones():
    """svgp_mnist_decoder_fwd_pre / _bwd_data_pre read the effective (parity-class) weights of the up-convolutions from ws.dec_weff,
    written once per step by a rider workgroup of svgp_mnist_encoder_kernel_matrix_fwd, instead of rebuilding them in every
    workgroup: bit-equal outputs; and the buffer follows the parameters (a changed decoder weight changes it)."""
    from svgp_vae_amd import _lib
    eng, _ = _engine(256, 32, 16)
    eng.run(adam=False)
    eng.synchronize()
    cfg, th, ws, st = C.byref(eng.cfg), eng.theta.data_ptr(), eng.ws.data_ptr(), eng.state.data_ptr()
    img, ax, s = eng._bound[0].data_ptr(), eng._bound[1].data_ptr(), eng.stream.cuda_stream
    fields = dict(dec_h0=(256, 128), dec_a1=(256, 512), dec_a2=(256, 1568), recon=(256, 784), zbar=(256, 16), dec_d2=(256, 1568),
                  dec_d1=(256, 512), dec_dh0=(256, 128))
    _lib.call("svgp_mnist_decoder_fwd", cfg, th, img, ws, s)
    _lib.call("svgp_mnist_decoder_bwd_data", cfg, th, img, ws, st, s)
    eng.synchronize()
    ref = {k: eng.ws_view(k, sh).clone() for k, sh in fields.items()}
    weff0 = eng.ws_view("dec_weff", (2176,)).clone()
    assert float(weff0.abs().max()) > 0
    for k, sh in fields.items():
        eng.ws_view(k, sh).fill_(float("nan"))
    _lib.call("svgp_mnist_decoder_fwd_pre", cfg, th, img, ws, s)
    _lib.call("svgp_mnist_decoder_bwd_data_pre", cfg, th, img, ws, st, s)
    eng.synchronize()
    for k, sh in fields.items():
        assert torch.equal(eng.ws_view(k, sh), ref[k]), k
    with torch.cuda.stream(eng.stream):
        eng.params["dec_c2_w"][1, 1, 3, 2] += 0.5
    _lib.call("svgp_mnist_encoder_kernel_matrix_fwd", cfg, th, img, ax, ws, s)
    eng.synchronize()
    assert not torch.equal(eng.ws_view("dec_weff", (2176,)), weff0)


@pytest.mark.parametrize("b,m,L,M", [(256, 32, 16, 8), (48, 16, 4, 4), (300, 24, 5, 4)])
def test_pass_two_of_the_reverse_row_stage_rides_in_the_encoder_reverse_launch(b, m, L, M):
    """svgp_gp_posterior_bwd_rows + svgp_mnist_encoder_bwd_km_sum == svgp_gp_posterior_bwd_with_final +
    svgp_kernel_matrix_bwd_partials + svgp_mnist_encoder_bwd, bit for bit: the channel sums (Knbar, knnbar, Kbar) are formed by
    workgroups at the head of the launch, the kernel-matrix VJP workgroups wait for them on the counter in ws.flags (agent-scope
    release / acquire), which every call leaves zero, with the error word clear; repeated calls stay equal."""
    from svgp_vae_amd import _lib
    params, images, aux, eps = H.toy_problem(b=b, m=m, L=L, M=M, n_obj=20, seed=6)
    eng = H.engine_for(params, b, geco=True, N_train=400.0)
    dev = eng.device
    eng.bind(images.to(dev), aux.to(dev), eps.to(dev))
    eng.run(adam=False)
    eng.synchronize()
    cfg, th, ws, st = C.byref(eng.cfg), eng.theta.data_ptr(), eng.ws.data_ptr(), eng.state.data_ptr()
    img, ax, s = eng._bound[0].data_ptr(), eng._bound[1].data_ptr(), eng.stream.cuda_stream
    nrb = (b + (256 // m) - 1) // (256 // m)
    views = lambda: dict(part_enc=eng.ws_view("part_enc", (eng.wl.n_part, eng.pl.n_enc)), d_on=eng.ws_view("d_on", (b, M)),
                         part_gp=eng.ws_view("part_gp", (m + nrb, 2)), ip=eng.grads()["inducing_index_points"],
                         Knbar=eng.ws_view("Knbar", (b, m)), knnbar=eng.ws_view("knnbar", (b,)), Kbar=eng.ws_view("Kbar", (m, m)),
                         ybar=eng.ws_view("ybar", (b, L)), s2bar=eng.ws_view("s2bar", (b, L)))
    _lib.call("svgp_gp_posterior_bwd_with_final", cfg, ws, st, s)
    _lib.call("svgp_kernel_matrix_bwd_partials", cfg, th, ax, ws, s)
    _lib.call("svgp_mnist_encoder_bwd", cfg, th, img, ws, s)
    eng.synchronize()
    ref = {k: v.clone() for k, v in views().items()}
    flags = eng.ws_view("flags", (64,))
    for rep in range(3):
        for v in views().values():
            v.fill_(float("nan"))
        _lib.call("svgp_gp_posterior_bwd_rows", cfg, ws, st, s)
        _lib.call("svgp_mnist_encoder_bwd_km_sum", cfg, th, img, ax, ws, st, s)
        eng.synchronize()
        assert torch.count_nonzero(flags.view(torch.int64)) == 0, (rep, flags.view(torch.int64).tolist())
        for k, v in views().items():
            assert torch.equal(v, ref[k]), (rep, k)


@pytest.mark.parametrize("four", [False, True])
@pytest.mark.parametrize("b,m,L", [(256, 32, 16), (48, 16, 4), (300, 24, 5), (7, 12, 3), (96, 48, 3)])
def test_reverse_statistics_ride_in_the_reverse_factor_launch(b, m, L, four, monkeypatch):
    """svgp_gp_stats_factor_bwd_wgrad == svgp_gp_stats_bwd + svgp_gp_factor_bwd_nofinal_wgrad, bit for bit: the P L statistics
    workgroups at the head of the launch write the row partials of A2 / ud / td through, channel workgroup l waits for its own P
    producers on ws.flags[8 + l] and re-arms it; every call leaves the counters zero and the error word clear.  Two forms of the
    channel workgroup: five LDS matrices (m <= 32: Ki stays resident, the wait sits behind Kbar_l = Abar G^T) and four (32 < m <= 64
    or SVGP_STAT_FOUR=1: the wait sits before T1 A)."""
    from svgp_vae_amd import _lib
    if four:
        monkeypatch.setenv("SVGP_STAT_FOUR", "1")
    eng, _ = _engine(b, m, L, seed=8)
    eng.run(adam=False)
    eng.synchronize()
    cfg, ws, st = C.byref(eng.cfg), eng.ws.data_ptr(), eng.state.data_ptr()
    img, s = eng._bound[0].data_ptr(), eng.stream.cuda_stream
    P = eng.wl.statB_len // (L * (m * m + 2 * m))
    n_dec = eng.pl.n_vae - eng.pl.n_enc
    shapes = dict(A2=(P, L, m, m), ud=(P, L, m), td=(P, L, m), fb_part=(2, L, m, m), vbar=(L, m), Ssym=(L, m, m), Qm=(L, m, m),
                  part_dec=(eng.wl.n_part, n_dec))
    _lib.call("svgp_gp_stats_bwd", cfg, ws, st, s)
    _lib.call("svgp_gp_factor_bwd_nofinal_wgrad", cfg, img, ws, st, s)
    eng.synchronize()
    ref = {k: eng.ws_view(k, sh).clone() for k, sh in shapes.items()}
    flags = eng.ws_view("flags", (64,))
    for rep in range(3):
        for k, sh in shapes.items():
            eng.ws_view(k, sh).fill_(float("nan"))
        _lib.call("svgp_gp_stats_factor_bwd_wgrad", cfg, img, ws, st, s)
        eng.synchronize()
        assert torch.count_nonzero(flags.view(torch.int64)) == 0, (rep, flags.view(torch.int64).tolist())
        for k, sh in shapes.items():
            assert torch.equal(eng.ws_view(k, sh), ref[k]), (rep, k)


def test_step_with_and_without_the_statistics_merge_is_bit_equal(monkeypatch):
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("SVGP_STAT_MERGE", flag)
        eng, _ = _engine(256, 32, 16, seed=9)
        for _ in range(3):
            eng.run(adam=True)
        eng.synchronize()
        out[flag] = ({k: v.clone() for k, v in eng.grads().items()}, eng.scalars()["elbo"], eng.theta.clone())
    assert out["0"][1] == out["1"][1]
    assert torch.equal(out["0"][2], out["1"][2])
    for k, v in out["0"][0].items():
        assert torch.equal(v, out["1"][0][k]), k


def test_merged_encoder_reverse_launch_keeps_two_workgroups_per_cu():
    """No launch-bounds occupancy hint on k_encoder_bwd_km (it slowed the image code by 1.7 us): the config-2 instance must stay at
    <= 168 registers per lane on its own, or a kernel-matrix VJP workgroup and an image workgroup no longer share a CU."""
    from svgp_vae_amd import _lib
    n = C.c_int(0)
    _lib.call("svgp_mnist_encoder_bwd_km_regs", C.byref(n))
    assert 0 < n.value <= 168, n.value


@pytest.mark.parametrize("b,m,L", [(256, 32, 16), (48, 16, 4), (300, 24, 5), (7, 12, 3)])
def test_deferred_inverse_rides_in_the_decoder_data_reverse_launch(b, m, L):
    """svgp_gp_posterior_fwd + svgp_mnist_decoder_bwd_data_pre_aji == svgp_gp_posterior_fwd_with_aji + svgp_mnist_decoder_bwd_data_pre,
    bit for bit ((A_hat + jI)^-1, the KL terms incl. -log det / 2, the row-stage outputs, zbar and the stored pre-activation
    gradients); and the kernel stays at <= 168 registers per lane (a rider and an image workgroup on one CU)."""
    from svgp_vae_amd import _lib
    eng, _ = _engine(b, m, L, seed=12)
    eng.run(adam=False)
    eng.synchronize()
    cfg, th, ws, st = C.byref(eng.cfg), eng.theta.data_ptr(), eng.ws.data_ptr(), eng.state.data_ptr()
    img, ep, s = eng._bound[0].data_ptr(), eng._bound[2].data_ptr(), eng.stream.cuda_stream
    shapes = dict(Aji=(L, m, m), KL=(2 * L,), p_m=(b, L), p_v=(b, L), z=(b, L), zbar=(b, L), dec_d2=(b, 1568), dec_d1=(b, 512),
                  dec_dh0=(b, 128))
    out = {}
    for form in ("row", "dec"):
        for k in ("Aji", "zbar", "dec_d2", "dec_d1", "dec_dh0"):
            eng.ws_view(k, shapes[k]).fill_(float("nan"))
        _lib.call("svgp_gp_factor_fwd_defer_aji", cfg, ws, s)
        if form == "row":
            _lib.call("svgp_gp_posterior_fwd_with_aji", cfg, ep, ws, st, s)
            _lib.call("svgp_mnist_decoder_fwd_pre", cfg, th, img, ws, s)
            _lib.call("svgp_mnist_decoder_bwd_data_pre", cfg, th, img, ws, st, s)
        else:
            _lib.call("svgp_gp_posterior_fwd", cfg, ep, ws, st, s)
            _lib.call("svgp_mnist_decoder_fwd_pre", cfg, th, img, ws, s)
            _lib.call("svgp_mnist_decoder_bwd_data_pre_aji", cfg, th, img, ws, st, s)
        eng.synchronize()
        out[form] = {k: eng.ws_view(k, sh).clone() for k, sh in shapes.items()}
    for k in shapes:
        assert torch.isfinite(out["dec"][k]).all(), k
        assert torch.equal(out["dec"][k], out["row"][k]), k
    n = C.c_int(0)
    _lib.call("svgp_mnist_decoder_bwd_data_aji_regs", C.byref(n))
    assert 0 < n.value <= 168, n.value


def test_step_with_and_without_the_inverse_in_the_decoder_launch_is_bit_equal(monkeypatch):
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("SVGP_AJI_DEC", flag)
        eng, _ = _engine(256, 32, 16, seed=13)
        for _ in range(3):
            eng.run(adam=True)
        eng.synchronize()
        out[flag] = (eng.theta.clone(), eng.state.clone(), eng.scalars()["elbo"])
    assert torch.equal(out["0"][0], out["1"][0]) and torch.equal(out["0"][1], out["1"][1]) and out["0"][2] == out["1"][2]


def test_timed_out_handoff_is_raised_not_returned():
    """A consumer workgroup that gives up waiting for its producer sets ws.flags[2]; the engine turns that into an error at the next
    read of the step's scalars (and re-arms the counters) instead of handing out the numbers of a broken step."""
    from svgp_vae_amd import _lib
    eng, _ = _engine(48, 16, 4, seed=18)
    eng.run(adam=False)
    assert "elbo" in eng.scalars()
    eng.ws_view("flags", (64,)).view(torch.int64)[2] = 1
    with pytest.raises(_lib.SvgpError, match="hand-off"):
        eng.scalars()
    assert torch.count_nonzero(eng.ws_view("flags", (64,)).view(torch.int64)) == 0
    eng.run(adam=False)
    assert "elbo" in eng.scalars()
