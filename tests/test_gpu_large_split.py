"""m > 64: the split stages of the two-stream step equal the unsplit ones bit for bit.

svgp_gp_factor_fwd == svgp_gp_factor_fwd_defer_aji + svgp_gp_factor_fwd_aji_tail, and
svgp_gp_factor_bwd == svgp_gp_factor_bwd_early + svgp_gp_factor_bwd_late (include/svgpvae_hip.h), with the tail / early half on
the caller's stream or on a second stream ordered by events -- the form svgp_mnist_step_phase and sprites.py use.
A whole step with SVGP_SIDE_STREAMS=0 (one stream) equals the default two-stream step.
"""
import ctypes as C
import os
import subprocess
import sys

import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

FWD_FIELDS = ("Ki", "ldK", "Si", "t", "G", "A", "Aji", "mu_hat", "u", "KL", "q")
BWD_FIELDS = ("Kbar", "vbar", "Ssym")


def _engine():
    p, images, aux, eps = H.toy_problem(b=150, m=130, L=3, M=24, n_obj=40, seed=4)
    eng = H.engine_for(p, 150, geco=True, N_train=500.0, jitter=1e-4)
    dev = eng.device
    eng.bind(images.to(dev), aux.to(dev), eps.to(dev))
    eng.run(adam=False)
    eng.synchronize()
    return eng


def _field(eng, name):
    off = getattr(eng.wl, name)
    m, L = eng.cfg.m, eng.cfg.L
    n = {"Ki": m * m, "ldK": 1, "Kbar": m * m, "KL": L, "q": eng.cfg.b, "t": L * m, "mu_hat": L * m, "u": L * m,
         "vbar": L * m}.get(name, L * m * m)
    return eng.ws[off:off + n].clone()


@pytest.mark.parametrize("two_streams", [False, True])
def test_split_factor_stages_equal_the_unsplit_ones(two_streams):
    from svgp_vae_amd import _lib
    eng = _engine()
    cfg, ws, st = C.byref(eng.cfg), eng.ws.data_ptr(), eng.state.data_ptr()
    main = eng.stream
    side = torch.cuda.Stream(device=eng.device) if two_streams else main
    ws0 = eng.ws.clone()                         # workspace after a full step: every stage input is valid

    def restore():
        main.synchronize(); side.synchronize()
        eng.ws.copy_(ws0)
        torch.cuda.synchronize()

    # ---- forward factor stage
    restore()
    _lib.call("svgp_gp_factor_fwd", cfg, ws, main.cuda_stream)
    main.synchronize()
    want = {k: _field(eng, k) for k in FWD_FIELDS}
    restore()
    _lib.call("svgp_gp_factor_fwd_defer_aji", cfg, ws, main.cuda_stream)
    side.wait_stream(main)
    _lib.call("svgp_gp_factor_fwd_aji_tail", cfg, ws, side.cuda_stream)
    main.wait_stream(side)
    main.synchronize()
    for k in FWD_FIELDS:
        assert torch.equal(_field(eng, k), want[k]), k

    # ---- reverse factor stage (inputs: the forward stage above + the step's reverse statistics)
    _lib.call("svgp_gp_factor_bwd", cfg, ws, st, main.cuda_stream)
    main.synchronize()
    wantb = {k: _field(eng, k) for k in BWD_FIELDS}
    ws1 = eng.ws.clone()
    for k in BWD_FIELDS:                          # poison the outputs, keep every input
        off = getattr(eng.wl, k)
        eng.ws[off:off + wantb[k].numel()].fill_(float("nan"))
    torch.cuda.synchronize()
    side.wait_stream(main)
    _lib.call("svgp_gp_factor_bwd_early", cfg, ws, st, side.cuda_stream)
    main.wait_stream(side)
    _lib.call("svgp_gp_factor_bwd_late", cfg, ws, st, main.cuda_stream)
    main.synchronize()
    for k in BWD_FIELDS:
        assert torch.equal(_field(eng, k), wantb[k]), k
    del ws1


def test_exchanged_symmetric_blocks_are_exactly_symmetric_in_memory():
    """the packed exchange moves the LOWER tiles of S, Sigma^-1, A2 and Ssym and the owner keeps its own window as it
    is -- owner and peers agree only because these blocks are exactly symmetric in memory (mirrored product stores, potri /
    k_symmetrize, the tile-pair kernel).  Pinned here, bit for bit, on the workspace of a full step (Cholesky route, m >= 512,
    and Gauss-Jordan route)."""
    for m, M in ((520, 40), (130, 24)):
        p, images, aux, eps = H.toy_problem(b=96, m=m, L=2, M=M, n_obj=40, seed=6)
        eng = H.engine_for(p, 96, geco=True, N_train=500.0, jitter=1e-2)
        dev = eng.device
        eng.bind(images.to(dev), aux.to(dev), eps.to(dev))
        eng.run(adam=False)
        eng.synchronize()
        bad = []
        for name in ("S", "Si", "A2", "Ssym", "A", "Aji"):
            X = eng.ws_view(name, (2, m, m))
            if not torch.equal(X, X.transpose(1, 2)):
                bad.append((m, name, float((X - X.transpose(1, 2)).abs().max())))
        Ki = eng.ws_view("Ki", (m, m))
        if not torch.equal(Ki, Ki.T):
            bad.append((m, "Ki", float((Ki - Ki.T).abs().max())))
        assert not bad, bad


def test_split_entry_points_reject_the_lds_path():
    from svgp_vae_amd import _lib
    p, images, aux, eps = H.toy_problem(b=40, m=12, L=3, M=4, n_obj=20, seed=0)
    eng = H.engine_for(p, 40, geco=False, N_train=300.0)
    for sym, args in (("svgp_gp_factor_fwd_aji_tail", (eng.ws.data_ptr(),)),
                      ("svgp_gp_factor_bwd_early", (eng.ws.data_ptr(), eng.state.data_ptr())),
                      ("svgp_gp_factor_bwd_late", (eng.ws.data_ptr(), eng.state.data_ptr()))):
        with pytest.raises(_lib.SvgpError):
            _lib.call(sym, C.byref(eng.cfg), *args, eng.stream.cuda_stream)


_CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from tests import helpers as H
tit = sys.argv[3] == "1"
# Titsias: enough channels that the batched inverse of the forward tail (side stream, if it were forked) and the Titsias
# inverse (main stream) would really be in flight together -- both go through ws.scr_inv
p, images, aux, eps = H.toy_problem(b=150, m=130, L=16 if tit else 3, M=24, n_obj=40, seed=4)
eng = H.engine_for(p, 150, geco=True, N_train=500.0, jitter=1e-4, titsias=tit)
dev = eng.device
eng.bind(images.to(dev), aux.to(dev), eps.to(dev))
for _ in range(2):
    eng.run(adam=True)
eng.synchronize()
torch.save({"theta": eng.theta.cpu(), "state": eng.state.cpu()}, sys.argv[2])
"""


@pytest.mark.parametrize("titsias", [False, True])
def test_two_stream_step_equals_one_stream_step(tmp_path, titsias):
    """Two Adam steps with the default schedule and with SVGP_SIDE_STREAMS=0: identical parameters and state
    (child processes: the switch is read from the environment of the library).  With cfg.titsias the default schedule
    must not fork at all (the Titsias stage and the forward tail share the inverse scratch)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for val in ("", "0"):
        out = tmp_path / f"s{val or 'default'}.pt"
        env = dict(os.environ)
        env.pop("SVGP_SIDE_STREAMS", None)
        if val:
            env["SVGP_SIDE_STREAMS"] = val
        subprocess.run([sys.executable, "-c", _CHILD, root, str(out), str(int(titsias))], check=True, env=env,
                       timeout=600)
        outs.append(torch.load(out))
    assert torch.equal(outs[0]["theta"], outs[1]["theta"])
    assert torch.equal(outs[0]["state"], outs[1]["state"])


def _run_stages(eng, order, images, eps):
    """Phase 1 + the reverse factor stage through the individual entry points, everything on ONE stream, with the work of
    the side branch (tail + early reverse half) issued either before or after the main-branch stages it runs beside."""
    from svgp_vae_amd import _lib
    cfg, ws, st, s = C.byref(eng.cfg), eng.ws.data_ptr(), eng.state.data_ptr(), eng.stream.cuda_stream
    th, im = eng.theta.data_ptr(), images.data_ptr()

    def side():
        _lib.call("svgp_gp_factor_fwd_aji_tail", cfg, ws, s)
        _lib.call("svgp_gp_factor_bwd_early", cfg, ws, st, s)

    def main():
        _lib.call("svgp_gp_posterior_fwd", cfg, eps.data_ptr(), ws, st, s)
        _lib.call("svgp_mnist_decoder_fwd", cfg, th, im, ws, s)
        _lib.call("svgp_mnist_decoder_bwd", cfg, th, im, ws, st, s)
        _lib.call("svgp_gp_stats_bwd", cfg, ws, st, s)

    _lib.call("svgp_gp_factor_fwd_defer_aji", cfg, ws, s)
    for part in ((side, main) if order == "side_first" else (main, side)):
        part()
    _lib.call("svgp_gp_factor_bwd_late", cfg, ws, st, s)
    _lib.call("svgp_gp_posterior_bwd", cfg, ws, st, s)
    eng.stream.synchronize()
    return eng.ws.clone()


def test_side_branch_and_main_branch_share_no_buffer():
    """The branch that svgp_mnist_step_phase runs on the side stream (forward tail + early reverse half) and the stages
    it runs beside (row stage, decoder forward / reverse, reverse statistics) are issued in both serial orders on one
    stream.  If either branch wrote a buffer the other reads or writes, one of the two orders would see different inputs:
    the whole workspace after the reverse row stage must be bit-identical (deterministic form of the overlap check)."""
    p, images, aux, eps = H.toy_problem(b=150, m=130, L=5, M=24, n_obj=40, seed=4)
    eng = H.engine_for(p, 150, geco=True, N_train=500.0, jitter=1e-4)
    dev = eng.device
    images, aux, eps = images.to(dev), aux.to(dev), eps.to(dev)
    eng.bind(images, aux, eps)
    eng.phase(0)
    eng.synchronize()
    ws0 = eng.ws.clone()
    out = {}
    for order in ("side_first", "main_first"):
        eng.ws.copy_(ws0)
        # scratch regions are poisoned differently per order, so that a stage reading scratch it has not written shows
        for name, n in (("scr_mm", 4 * eng.cfg.L * 130 * 130), ("fb_part", 2 * eng.cfg.L * 130 * 130)):
            off = getattr(eng.wl, name)
            eng.ws[off:off + n].fill_(1e300 if order == "side_first" else -3.0)
        torch.cuda.synchronize()
        out[order] = _run_stages(eng, order, images, eps)
    a, b = out["side_first"], out["main_first"]
    outputs = ("Kbar", "vbar", "Ssym", "Aji", "KL", "Knbar", "knnbar", "ybar", "s2bar", "p_m", "p_v", "z", "zbar", "A2", "ud",
               "td", "Qm")
    for k in outputs:
        off = getattr(eng.wl, k)
        nxt = min([getattr(eng.wl, f) for f, _ in eng.wl._fields_ if getattr(eng.wl, f) > off
                   and f not in ("total", "n_part", "n_post", "stat_parts") and not f.endswith("_len")] + [a.numel()])
        assert torch.equal(a[off:nxt], b[off:nxt]), k
        assert torch.isfinite(a[off:nxt]).all(), k


@pytest.mark.parametrize("m,b,L,M", [(130, 150, 3, 24), (256, 1024, 16, 32)])
def test_k_only_side_branch_equals_the_in_line_order(m, b, L, M, monkeypatch):
    """the channel-independent block of the forward factor stage ((K + jI)^-1, Kn Ki, q, W, P^T) on side branch 1
    beside the statistics and the channel inverses (SVGP_KONLY_BRANCH, default on for 64 < m < 512) against the in-line order:
    the same operations on the same values, so two Adam steps agree BIT FOR BIT -- which also pins that the branch's scratch
    (the tail of the inverse workspace, Kn Ki / W / P^T) is disjoint from what the statistics and the channel block use.
    Second shape: BASELINE configs[2]."""
    params, images, aux, eps = H.toy_problem(b=b, m=m, L=L, M=M, n_obj=40, seed=9)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SVGP_KONLY_BRANCH", flag)
        eng = H.engine_for(params, b, geco=True, N_train=4050.0, jitter=1e-4)
        dev = eng.device
        eng.bind(images.to(dev), aux.to(dev), eps.to(dev))
        for _ in range(2):
            eng.run(adam=True)
        eng.synchronize()
        res[flag] = (eng.theta.clone(), eng.state.clone())
    assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1])


@pytest.mark.parametrize("m,b,L,M", [(130, 150, 3, 24), (256, 1024, 16, 32)])
def test_ki_gradient_chain_on_the_side_branch_equals_the_one_stream_order(m, b, L, M, monkeypatch):
    """Round 6: the single-matrix chain of the gradient of Ki (svgp_gp_factor_bwd_late_b_kbar) runs on side branch 1 beside the channel
    block of the late reverse factor half (SVGP_KBAR_BRANCH, default on): the same operations on the same values as the one-stream
    order, so two Adam steps agree BIT FOR BIT -- which also pins that the chain's scratch (tA, tB, tC, Pbar) is disjoint from the
    channel block's.  Second shape: BASELINE configs[2]."""
    params, images, aux, eps = H.toy_problem(b=b, m=m, L=L, M=M, n_obj=40, seed=10)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SVGP_KBAR_BRANCH", flag)
        eng = H.engine_for(params, b, geco=True, N_train=4050.0, jitter=1e-4)
        dev = eng.device
        eng.bind(images.to(dev), aux.to(dev), eps.to(dev))
        for _ in range(2):
            eng.run(adam=True)
        eng.synchronize()
        res[flag] = (eng.theta.clone(), eng.state.clone())
    assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1])


def test_late_reverse_half_in_three_pieces_equals_the_one_call():
    """svgp_gp_factor_bwd_late_b == _late_b_channels; _late_b_kbar; _late_b_final (and with the two independent pieces swapped)."""
    from svgp_vae_amd import _lib
    eng = _engine()
    cfg, ws, st = C.byref(eng.cfg), eng.ws.data_ptr(), eng.state.data_ptr()
    s = eng.stream.cuda_stream
    _lib.call("svgp_gp_factor_bwd_early", cfg, ws, st, s)
    _lib.call("svgp_gp_factor_bwd_late_a", cfg, ws, st, s)
    eng.synchronize()
    ws0 = eng.ws.clone()
    _lib.call("svgp_gp_factor_bwd_late_b", cfg, ws, st, s)
    eng.synchronize()
    want = {k: _field(eng, k) for k in BWD_FIELDS}
    for order in (("channels", "kbar"), ("kbar", "channels")):
        eng.ws.copy_(ws0)
        torch.cuda.synchronize()
        for piece in order + ("final",):
            _lib.call("svgp_gp_factor_bwd_late_b_" + piece, cfg, ws, st, s)
        eng.synchronize()
        for k in BWD_FIELDS:
            assert torch.equal(_field(eng, k), want[k]), (order, k)
