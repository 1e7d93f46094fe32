"""CPU-side checks of the drop-in boundary: the built library loads, exports every symbol
include/svgpvae_hip.h declares, the ctypes binding covers them all, layouts are consistent, and
the product path fails loudly without a GPU / without the extension (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import svgp_vae_amd
from svgp_vae_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "svgpvae_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svgp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(svgp_vae_amd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = C.CDLL(svgp_vae_amd.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/svgpvae_hip.h but not exported"
    bound = set(_lib.SIGNATURES) | set(_lib.NON_STATUS)
    assert bound == set(names), (sorted(bound - set(names)), sorted(set(names) - bound))


def test_layouts_match_reference_parameter_count_and_are_disjoint():
    lib = svgp_vae_amd.load_library()
    assert lib.svgp_version() == 1
    cfg = _lib.MnistCfg(b=256, b_global=256, m=32, L=16, M=8, n_obj=400, N_train=4050.0, jitter=1e-6)
    pl, wl = _lib.ParamLayout(), _lib.WsLayout()
    _lib.call("svgp_mnist_param_layout_get", C.byref(cfg), C.byref(pl))
    _lib.call("svgp_mnist_ws_layout_get", C.byref(cfg), C.byref(wl))
    # 5721 VAE + 32x10 inducing + l, amp + 400x8 object vectors = 9243 trainables
    assert (pl.n_enc, pl.n_vae, pl.n_total) == (2304, 5721, 9243)
    tit = ("tit_S2", "tit_v2", "tit_Si", "tit_t", "tit_scal",     # zero-sized unless cfg.titsias
           "xpack", "xpack_len",                                  # zero-sized unless m > 64 and sharded over ranks
           "scr_sm")                                              # zero-sized unless m > 64
    offs = sorted(getattr(wl, f) for f in _lib.WS_FIELDS
                  if f not in ("statA_len", "statB_len", "gradC_len", "n_part", "n_post", "total",
                               "statA", "statB", "gradC") + tit)
    assert all(o >= 0 for o in offs) and offs[-1] < wl.total
    # Titsias: S2 | v2 extend the statA exchange block, the rest sits at the end
    cfg_t = _lib.MnistCfg(b=256, b_global=256, m=32, L=16, M=8, n_obj=400, N_train=4050.0, jitter=1e-6, titsias=1)
    wt = _lib.WsLayout()
    _lib.call("svgp_mnist_ws_layout_get", C.byref(cfg_t), C.byref(wt))
    P = wl.stat_parts                       # row partials of the statistics blocks (b_cap >= 128, m <= 64)
    assert P == 4 and wt.stat_parts == 4
    assert wt.statA_len == (P + 1) * 16 * 32 * 33 and wt.tit_S2 == wt.v + P * 16 * 32 and wt.tit_v2 == wt.tit_S2 + 16 * 32 * 32
    assert wt.tit_scal + 2 * 16 + 1 <= wt.total and wt.total > wl.total
    assert wl.statA == wl.S and wl.v == wl.S + P * 16 * 32 * 32 and wl.statA_len == P * 16 * 32 * 33
    assert wl.statB_len == P * 16 * 32 * 34 and wl.gradC_len == 9243 + 8 and wl.sums == wl.grad + 9243
    small = _lib.MnistCfg(b=64, b_global=64, m=32, L=16, M=8, n_obj=400, N_train=4050.0, jitter=1e-6)
    big_m = _lib.MnistCfg(b=256, b_global=256, m=72, L=16, M=16, n_obj=400, N_train=4050.0, jitter=1e-6)
    sharded = _lib.MnistCfg(b=256, b_global=512, m=32, L=16, M=8, n_obj=400, N_train=4050.0, jitter=1e-6, single_stat_block=1)
    for c_ in (small, big_m, sharded):      # few rows / the global-memory path / blocks that are all-reduced: one block
        w_ = _lib.WsLayout()
        _lib.call("svgp_mnist_ws_layout_get", C.byref(c_), C.byref(w_))
        assert w_.stat_parts == 1


def test_bad_shapes_are_rejected_with_a_message():
    cfg = _lib.MnistCfg(b=256, b_global=256, m=4096, L=16, M=8, n_obj=400, N_train=4050.0, jitter=1e-6)
    with pytest.raises(svgp_vae_amd.SvgpError, match="m <= 2048"):
        _lib.call("svgp_mnist_ws_layout_get", C.byref(cfg), C.byref(_lib.WsLayout()))
    cfg = _lib.MnistCfg(b=0, b_global=0, m=8, L=1, M=1, n_obj=0, N_train=1.0)
    with pytest.raises(svgp_vae_amd.SvgpError, match="b_global"):
        _lib.call("svgp_mnist_param_layout_get", C.byref(cfg), C.byref(_lib.ParamLayout()))


def test_null_pointers_are_rejected_before_any_launch():
    cfg = _lib.MnistCfg(b=4, b_global=4, m=8, L=2, M=2, n_obj=0, N_train=10.0, jitter=1e-6)
    with pytest.raises(svgp_vae_amd.SvgpError, match="NULL"):
        _lib.call("svgp_mnist_encoder_fwd", C.byref(cfg), None, None, None, None)
    with pytest.raises(svgp_vae_amd.SvgpError, match="phase"):
        _lib.call("svgp_mnist_step_phase", C.byref(cfg), 7, 1, 1, 1, None, 1, 1, None, None, None)


def test_newer_entry_points_validate_their_arguments():
    """Argument checks of the streaming / communicator / Titsias entry points happen before any device call."""
    fake = C.c_void_p(4096)                                     # never dereferenced: every case fails validation first
    kd = _lib.StreamKdesc(kind=0, d1=2, d2=5, normalize=0, n_table=0)       # GPLVM dim 5 has no float32 instantiation
    with pytest.raises(svgp_vae_amd.SvgpError, match="no instantiation"):
        _lib.call("svgp_stream_knm_f32", C.byref(kd), 10, 8, fake, fake, fake, None)
    kd_bad = _lib.StreamKdesc(kind=0, d1=3, d2=8)
    with pytest.raises(svgp_vae_amd.SvgpError, match="d1 = 2"):
        _lib.call("svgp_stream_features_f32", C.byref(kd_bad), 10, fake, 10, 0, None, fake, None)
    lib = svgp_vae_amd.load_library()
    need = lib.svgp_stream_stats_workspace_elems(1000, 64, 3)
    assert need > 0
    with pytest.raises(svgp_vae_amd.SvgpError, match="workspace"):
        _lib.call("svgp_stream_stats_f32", 1000, 64, 3, fake, fake, fake, fake, fake, fake, need - 1, None)
    with pytest.raises(svgp_vae_amd.SvgpError, match="unique id"):
        _lib.call("svgp_comm_init", fake, 7, 0, 1, C.byref(C.c_void_p()))
    cfg = _lib.MnistCfg(b=4, b_global=4, m=8, L=2, M=2, n_obj=0, N_train=10.0, jitter=1e-6)       # titsias = 0
    with pytest.raises(svgp_vae_amd.SvgpError, match="titsias"):
        _lib.call("svgp_gp_titsias_stats", C.byref(cfg), fake, None)
    assert lib.svgp_spd_inverse_workspace_elems(800, 4) > lib.svgp_spd_inverse_workspace_elems(400, 4)


def test_moving_ball_entry_points_validate_their_arguments():
    fake = C.c_void_p(4096)                                     # never dereferenced
    cfg = _lib.MnistCfg(b=30, b_global=30, m=80, L=35, M=1, n_obj=0, N_train=30.0, jitter=1e-6, kl_form=1, clip_pv=2)
    with pytest.raises(svgp_vae_amd.SvgpError, match="moving-ball"):
        _lib.call("svgp_mnist_ws_layout_get", C.byref(cfg), C.byref(_lib.WsLayout()))
    cfg = _lib.MnistCfg(b=30, b_global=30, m=15, L=35, M=1, n_obj=0, N_train=30.0, jitter=1e-6, clip_pv=3)
    with pytest.raises(svgp_vae_amd.SvgpError, match="clip_pv"):
        _lib.call("svgp_mnist_ws_layout_get", C.byref(cfg), C.byref(_lib.WsLayout()))
    # the ball configuration itself lays out: KL field holds [KL | traces]
    cfg = _lib.MnistCfg(b=30, b_global=30, m=15, L=35, M=1, n_obj=0, N_train=30.0, jitter=1e-9, kl_form=1, clip_pv=2, b_cap=30)
    wl = _lib.WsLayout()
    _lib.call("svgp_mnist_ws_layout_get", C.byref(cfg), C.byref(wl))
    assert wl.q - wl.KL >= 2 * 35
    q = _lib.PearceBufs(B=35, T=100, n=100)
    with pytest.raises(svgp_vae_amd.SvgpError, match="n <= 64"):
        _lib.call("svgp_pearce_gp_fwd", C.byref(q), None, None, fake, None)
    q = _lib.PearceBufs(B=35, T=30, n=12)                       # a context set needs its index list
    with pytest.raises(svgp_vae_amd.SvgpError, match="index set"):
        _lib.call("svgp_pearce_gp_fwd", C.byref(q), None, None, fake, None)
    with pytest.raises(svgp_vae_amd.SvgpError, match="NULL"):
        _lib.call("svgp_se1d_kernel_matrix_fwd", 30, 15, None, fake, fake, fake, fake, fake, None)
    with pytest.raises(svgp_vae_amd.SvgpError, match="slot"):
        _lib.call("svgp_state_add", fake, 99, 1.0, None)


def test_missing_extension_fails_loudly(tmp_path):
    with pytest.raises(svgp_vae_amd.SvgpError, match="no CPU fallback"):
        svgp_vae_amd.load_library(str(tmp_path / "nope.so"))


def test_engine_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from svgp_vae_amd.engine import MnistStepEngine
    with pytest.raises(svgp_vae_amd.SvgpError, match="no CPU execution path"):
        MnistStepEngine(32)
    from svgp_vae_amd import ball
    with pytest.raises(svgp_vae_amd.SvgpError, match="no CPU execution path"):
        ball.PearceStepEngine("VAE", 0.001, batch=4, tmax=8, px=8, py=8, hidden=8)
    mk = lambda n: ball.SVGP(False, 5, False, 1, 8, 2.0, False, n, 1e-6, 1, 8, 2.0)
    with pytest.raises(svgp_vae_amd.SvgpError, match="no CPU execution path"):
        ball.BallStepEngine(mk("x"), mk("y"), batch=4, tmax=8, px=8, py=8, hidden=8)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "svgp-vae_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn


def _header_struct_fields(name):
    """[(ctype, field)] of `typedef struct { ... } name;` in include/svgpvae_hip.h, in declaration order."""
    src = open(os.path.join(ROOT, "include", "svgpvae_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    body = re.search(r"typedef\s+struct\s*\{([^}]*)\}\s*" + name + r"\s*;", src, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        mm = re.match(r"(const\s+)?([A-Za-z_0-9]+)\s*(\*?)\s*(.*)$", decl, flags=re.S)
        ctype = mm.group(2) + mm.group(3)
        for fld in mm.group(4).split(","):
            fld = fld.strip().lstrip("*").strip()
            out.append((ctype, re.sub(r"\[.*\]", "", fld)))
    return out


_CT = {"int32_t": C.c_int32, "int64_t": C.c_int64, "double": C.c_double, "float": C.c_float}


def test_ctypes_mirrors_follow_the_header_field_for_field():
    """a stale mirror of svgp_mnist_cfg mis-aligns every double.  The header is the source of truth:
    _lib's structures, the INTEGRATION.md stub and the compiled library must agree with it."""
    for cname, cls in (("svgp_mnist_cfg", _lib.MnistCfg), ("svgp_mnist_param_layout", _lib.ParamLayout),
                       ("svgp_mnist_ws_layout", _lib.WsLayout), ("svgp_sprites_kcfg", _lib.SpritesKcfg)):
        want = _header_struct_fields(cname)
        got = [(n, t) for n, t in cls._fields_]
        assert [f for _, f in want] == [n for n, _ in got], cname
        assert [_CT[t] for t, _ in want] == [t for _, t in got], cname
    # array members: names + order only
    for cname, cls in (("svgp_conv_desc", _lib.ConvDesc), ("svgp_stream_kdesc", _lib.StreamKdesc),
                       ("svgp_pearce_bufs", _lib.PearceBufs), ("svgp_sum_job", _lib.SumJob)):
        assert [f for _, f in _header_struct_fields(cname)] == [n for n, _ in cls._fields_], cname
    lib = svgp_vae_amd.load_library()
    for which, cls in enumerate((_lib.MnistCfg, _lib.ParamLayout, _lib.WsLayout, _lib.StreamKdesc, _lib.ConvDesc,
                                 _lib.SpritesKcfg, _lib.PearceBufs, _lib.SumJob)):
        assert lib.svgp_struct_sizeof(which) == C.sizeof(cls), cls.__name__
    assert lib.svgp_struct_sizeof(99) == -1
    # the INTEGRATION.md stub: executed as written (minus the CDLL / call lines) it must define the same structure
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    snippet = re.search(r"(class MnistCfg\(C\.Structure\):.*?)\nlib\.svgp_struct_sizeof", doc, flags=re.S).group(1)
    ns = {"C": C}
    exec(snippet, ns)
    assert [(n, t) for n, t in ns["MnistCfg"]._fields_] == [(n, t) for n, t in _lib.MnistCfg._fields_]
    assert C.sizeof(ns["MnistCfg"]) == lib.svgp_struct_sizeof(0)
