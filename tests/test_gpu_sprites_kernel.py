"""spritesSVGP.kernel_matrix (SVGPVAE_model.py:550-600) as the C entry points svgp_sprites_kernel_matrix_fwd / _bwd against a float64
torch restatement + autograd: ragged sizes (nothing a multiple of the 16-target tiles / 64-source chunks), the three kernel kinds, and
feature groups in both template buckets of the tiled reverse pass, (8, 16) and (16, 32), and beyond them (per-target reference kernels)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DT = torch.float64


def _grp(kind, x, y, ell, sig):
    if kind == "se":
        d2 = ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1)
        return sig * sig * torch.exp(-d2 / (2 * ell * ell))
    xy = x @ y.T
    if kind == "lin":
        return xy
    return xy / (x.norm(dim=1)[:, None] * y.norm(dim=1)[None, :])


def _mats(kind, ids, chars, ip, table, se, La):
    xa = table[ids]
    Kn = _grp(kind, xa, ip[:, :La], se[0], se[1]) * _grp(kind, chars, ip[:, La:], se[2], se[3])
    K = _grp(kind, ip[:, :La], ip[:, :La], se[0], se[1]) * _grp(kind, ip[:, La:], ip[:, La:], se[2], se[3])
    knn = torch.stack([_grp(kind, xa[i:i + 1], xa[i:i + 1], se[0], se[1])[0, 0] * _grp(kind, chars[i:i + 1], chars[i:i + 1], se[2], se[3])[0, 0]
                       for i in range(len(ids))])
    return K, Kn, knn


@pytest.mark.parametrize("kind", ["cos", "lin", "se"])
@pytest.mark.parametrize("b,m,La,Lc,n_act", [(137, 203, 5, 11, 9), (70, 33, 8, 16, 4), (45, 90, 3, 20, 6), (33, 50, 12, 7, 5), (40, 37, 20, 9, 3),
                                              (29, 41, 16, 32, 4)])
def test_sprites_kernel_matrix_and_vjp(kind, b, m, La, Lc, n_act):
    _run_case(kind, b, m, La, Lc, n_act, cap=b)


@pytest.mark.parametrize("b,cap,m", [(720, 936, 800), (1152, 1224, 800), (100, 936, 800), (936, 936, 800)])
def test_scratch_sized_at_the_row_capacity_covers_smaller_batches(b, cap, m):
    """the scratch need is not monotone in b (the split count of the batch-row targets grows as b shrinks).  The size
    function takes the CAPACITY and must cover every b below it: the call runs with b < cap on a capacity-sized scratch whose
    guard words behind the end must stay untouched; an undersized scratch is refused before any launch."""
    _run_case("lin", b, m, 8, 16, 72, cap=cap)


def _run_case(kind, b, m, La, Lc, n_act, cap):
    from svgp_vae_amd import _lib
    from svgp_vae_amd._lib import SpritesKcfg, call
    g = torch.Generator().manual_seed(b + 7 * m + La)
    ids = torch.randint(0, n_act, (b,), generator=g)
    # (SE kernel: features scaled down so that the off-diagonal kernel values do not vanish at 48 features -- with every
    # off-diagonal entry ~1e-11 the analytically zero diagonal pairs' rounding residue, 1e-16, dominates the relative error)
    fs = 0.3 if kind == "se" else 1.0
    chars = fs * torch.randn(b, Lc, generator=g, dtype=DT)
    ip = fs * torch.randn(m, La + Lc, generator=g, dtype=DT)
    table = fs * torch.randn(n_act, La, generator=g, dtype=DT)
    se = torch.tensor([1.3, 0.7, 0.9, 1.1], dtype=DT)
    gK, gKn, gknn = torch.randn(m, m, generator=g, dtype=DT), torch.randn(b, m, generator=g, dtype=DT), torch.randn(b, generator=g, dtype=DT)
    rw = 0.75
    cr, ipr, tr, ser = (t.clone().requires_grad_() for t in (chars, ip, table, se))
    K, Kn, knn = _mats(kind, ids, cr, ipr, tr, ser, La)
    ((rw * gK * K).sum() + (gKn * Kn).sum() + (gknn * knn).sum()).backward()
    dev = "cuda"
    aux = torch.cat([ids.to(DT)[:, None], chars], 1).to(dev)
    d = {k: v.to(dev) for k, v in dict(ip=ip, table=table, se=se, gK=gK, gKn=gKn, gknn=gknn).items()}
    kc = SpritesKcfg(b=b, m=m, La=La, Lc=Lc, n_act=n_act, normalize=int(kind == "cos"), k_se=int(kind == "se"), rep_weight=rw)
    oK, oKn, oknn = (torch.full(s, float("nan"), dtype=DT, device=dev) for s in ((m, m), (b, m), (b,)))
    s = torch.cuda.current_stream().cuda_stream
    call("svgp_sprites_kernel_matrix_fwd", C.byref(kc), aux.data_ptr(), d["ip"].data_ptr(), d["table"].data_ptr(), d["se"].data_ptr(),
         oK.data_ptr(), oKn.data_ptr(), oknn.data_ptr(), s)
    kc_cap = SpritesKcfg(b=cap, m=m, La=La, Lc=Lc, n_act=n_act, normalize=int(kind == "cos"), k_se=int(kind == "se"), rep_weight=rw)
    n_scr = int(_lib.load_library().svgp_sprites_kernel_bwd_scratch_elems(C.byref(kc_cap)))
    assert n_scr >= int(_lib.load_library().svgp_sprites_kernel_bwd_scratch_elems(C.byref(kc)))
    GUARD = 4096
    scr = torch.full((n_scr + GUARD,), float("nan"), dtype=DT, device=dev)  # every partial that is summed must have been written
    scr[n_scr:] = 12345.0
    d_ip, d_tab, d_char, d_se = (torch.full(sh, float("nan"), dtype=DT, device=dev) for sh in ((m, La + Lc), (n_act, La), (b, Lc), (4,)))
    call("svgp_sprites_kernel_matrix_bwd", C.byref(kc), aux.data_ptr(), d["ip"].data_ptr(), d["table"].data_ptr(), d["se"].data_ptr(),
         d["gK"].data_ptr(), d["gKn"].data_ptr(), d["gknn"].data_ptr(), d_ip.data_ptr(), d_tab.data_ptr(), d_char.data_ptr(),
         d_se.data_ptr(), scr.data_ptr(), n_scr, s)
    torch.cuda.synchronize()
    assert bool((scr[n_scr:] == 12345.0).all()), "the reverse pass wrote behind the end of its scratch"
    rc = _lib.load_library().svgp_sprites_kernel_matrix_bwd(
        C.byref(kc), aux.data_ptr(), d["ip"].data_ptr(), d["table"].data_ptr(), d["se"].data_ptr(), d["gK"].data_ptr(),
        d["gKn"].data_ptr(), d["gknn"].data_ptr(), d_ip.data_ptr(), d_tab.data_ptr(), d_char.data_ptr(), d_se.data_ptr(),
        scr.data_ptr(), 16, s)
    assert rc != 0, "an undersized scratch must be refused"
    rel = lambda a, c: float((a.cpu() - c).abs().max() / (c.abs().max() + 1e-300))
    assert rel(oK, K.detach()) < 1e-13 and rel(oKn, Kn.detach()) < 1e-13 and rel(oknn, knn.detach()) < 1e-13
    assert rel(d_ip, ipr.grad) < 1e-11 and rel(d_tab, tr.grad) < 1e-11 and rel(d_char, cr.grad) < 1e-11
    if kind == "se":
        assert rel(d_se, ser.grad) < 1e-11
    else:
        assert float(d_se.abs().max()) == 0.0
