"""float32 streaming statistics (svgp_stream_*): K_nm build and S_l / v_l against a float64 restatement of
SVGPVAE_model.py:427-476 (mnistSVGP.kernel_matrix), :550-600 (spritesSVGP.kernel_matrix) and :1004-1017
(precompute_GP_params_SVGPVAE) on the same inputs.  Tolerance: float32 arithmetic against a float64 oracle --
K_nm 2e-5 of max |K| (features, exp and products in fp32), statistics 2e-4 of max |S| (fp32 MFMA accumulation
over n rows); north_star allows 1e-3."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

K_TOL, S_TOL = 2e-5, 2e-4


def _mnist_kernel64(x, y, l_gp, amp, table, normalize, x_inducing):
    """float64 restatement (numpy) of mnistSVGP.kernel_matrix (SVGPVAE_model.py:427-476), y always inducing."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    ox = x[:, 2:] if (x_inducing or table is None) else np.asarray(table, np.float64)[x[:, 0].astype(int)]
    oy = y[:, 2:]
    d = x[:, 1][:, None] - y[:, 1][None, :]
    per = amp ** 2 * np.exp(-2.0 * np.sin(d / 2.0) ** 2 / l_gp ** 2)
    lin = ox @ oy.T
    if normalize:
        lin = lin / (np.linalg.norm(ox, axis=1)[:, None] * np.linalg.norm(oy, axis=1)[None, :])
    return per * lin


def _sprites_kernel64(x, y, table, normalize, se, x_inducing, La):
    """float64 restatement of spritesSVGP.kernel_matrix (:550-600), y always inducing."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    if x_inducing:
        ax, cx = x[:, :La], x[:, La:]
    else:
        ax, cx = np.asarray(table, np.float64)[x[:, 0].astype(int)], x[:, 1:]
    ay, cy = y[:, :La], y[:, La:]
    if se is not None:
        l1, s1, l2, s2 = se
        d1 = ((ax[:, None, :] - ay[None, :, :]) ** 2).sum(-1)
        d2 = ((cx[:, None, :] - cy[None, :, :]) ** 2).sum(-1)
        return s1 ** 2 * np.exp(-d1 / (2 * l1 ** 2)) * s2 ** 2 * np.exp(-d2 / (2 * l2 ** 2))
    k1, k2 = ax @ ay.T, cx @ cy.T
    if normalize:
        k1 = k1 / (np.linalg.norm(ax, axis=1)[:, None] * np.linalg.norm(ay, axis=1)[None, :])
        k2 = k2 / (np.linalg.norm(cx, axis=1)[:, None] * np.linalg.norm(cy, axis=1)[None, :])
    return k1 * k2


def _stats64(K, means, vars_):
    K, means, vars_ = (np.asarray(a, np.float64) for a in (K, means, vars_))
    p = np.where(vars_ == 0, 0.0, 1.0 / np.where(vars_ == 0, 1.0, vars_))
    S = np.einsum("ni,nl,nj->lij", K, p, K)
    v = np.einsum("ni,nl->li", K, p * means)
    return S, v


def _cuda(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda:0").contiguous()


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())


@pytest.mark.parametrize("n,m,M,table,normalize", [
    (300, 64, 8, True, False),     # rows not a multiple of the 64-row block
    (129, 52, 8, False, True),     # m % 4 == 0 but not a multiple of anything larger; no table; cosine
    (70, 37, 8, True, False),      # m % 4 != 0: scalar store path
    (257, 1100, 32, True, True),   # two column panels, M = 32 (config 3's GPLVM dim)
])
def test_knm_periodic_linear(n, m, M, table, normalize):
    from svgp_vae_amd import stream_stats as SS
    rng = np.random.default_rng(n + m)
    n_obj = 40
    tab = rng.normal(0, 1.5, (n_obj, M))
    x = np.concatenate([rng.integers(0, n_obj, (n, 1)).astype(float), rng.uniform(0, 2 * np.pi, (n, 1)),
                        rng.normal(0, 1.5, (n, M))], 1)
    z = np.concatenate([np.zeros((m, 1)), rng.uniform(0, 2 * np.pi, (m, 1)), rng.normal(0, 1.5, (m, M))], 1)
    l_gp, amp = 1.3, 0.9
    kd = SS.kernel_desc(SS.PERIODIC_LINEAR, 2, M, normalize=normalize, n_table=n_obj if table else 0, params=(l_gp, amp))
    dt = _cuda(tab) if table else None
    fr = SS.features(kd, _cuda(x), inducing=False, table=dt)
    fi = SS.features(kd, _cuda(z), inducing=True)
    K = SS.knm(kd, fr, n, fi, m).cpu().numpy()
    ref = _mnist_kernel64(x, z, l_gp, amp, tab if table else None, normalize, False)
    assert _rel(K, ref) < K_TOL
    Kmm = SS.knm(kd, fi, m, fi, m).cpu().numpy()
    assert _rel(Kmm, _mnist_kernel64(z, z, l_gp, amp, None, normalize, True)) < K_TOL


@pytest.mark.parametrize("se,normalize", [(None, False), (None, True), ((5.0, 1.4, 7.0, 1.2), False)])
def test_knm_sprites(se, normalize):
    from svgp_vae_amd import stream_stats as SS
    rng = np.random.default_rng(3)
    n, m, La, Lc = 333, 200, 8, 16
    tab = rng.normal(0, 1.5, (72, La))
    x = np.concatenate([rng.integers(0, 72, (n, 1)).astype(float), rng.normal(0, 1.5, (n, Lc))], 1)
    z = rng.normal(0, 1.5, (m, La + Lc))
    kd = SS.kernel_desc(SS.SE_SE if se else SS.LINEAR_LINEAR, La, Lc, normalize=normalize, n_table=72, params=se or ())
    fr = SS.features(kd, _cuda(x), inducing=False, table=_cuda(tab))
    fi = SS.features(kd, _cuda(z), inducing=True)
    K = SS.knm(kd, fr, n, fi, m).cpu().numpy()
    assert _rel(K, _sprites_kernel64(x, z, tab, normalize, se, False, La)) < K_TOL


@pytest.mark.parametrize("n,m,L", [
    (1000, 64, 3),      # one tile, one row slice, rows not a multiple of 16
    (5000, 300, 5),     # 2 x 2 tiles (one partial), mirrored lower tile, several row slices
    (4100, 516, 2),     # 3 x 3 tiles
    (40, 30, 17),       # fewer rows than a chunk; m % 4 != 0 (scalar loads); L > 16 (two channel passes for v)
])
def test_stats(n, m, L):
    from svgp_vae_amd import stream_stats as SS
    rng = np.random.default_rng(n)
    K = rng.normal(0, 1, (n, m)).astype(np.float32)
    means = rng.normal(0, 1, (n, L)).astype(np.float32)
    vars_ = rng.uniform(1e-3, 10, (n, L)).astype(np.float32)
    vars_[::7, 0] = 0.0                                       # reciprocal_no_nan rows
    S, v = SS.stats(_cuda(K), _cuda(means), _cuda(vars_))
    Sr, vr = _stats64(K, means, vars_)
    S, v = S.cpu().numpy(), v.cpu().numpy()
    assert _rel(S, Sr) < S_TOL
    assert _rel(v, vr) < S_TOL
    assert np.array_equal(S, S.transpose(0, 2, 1)[:, :, :]) or _rel(S, S.transpose(0, 2, 1).astype(np.float64)) < 1e-6


def test_precompute_gp_params_matches_float64():
    """precompute_GP_params_SVGPVAE end to end (K_nm, statistics, inverse, mean term) on a full-rank SE kernel."""
    from svgp_vae_amd import stream_stats as SS
    rng = np.random.default_rng(11)
    n, m, La, Lc, L = 3000, 96, 4, 6, 4
    se = (2.0, 1.0, 2.5, 1.0)
    tab = rng.normal(0, 1.0, (72, La))
    x = np.concatenate([rng.integers(0, 72, (n, 1)).astype(float), rng.normal(0, 1.0, (n, Lc))], 1)
    z = rng.normal(0, 1.0, (m, La + Lc))
    means = rng.normal(0, 1, (n, L)); vars_ = rng.uniform(0.5, 2.0, (n, L))
    kd = SS.kernel_desc(SS.SE_SE, La, Lc, n_table=72, params=se)
    mt, inv = SS.precompute_GP_params_f32(kd, _cuda(means), _cuda(vars_), _cuda(x), _cuda(z), table=_cuda(tab))
    Knm = _sprites_kernel64(x, z, tab, False, se, False, La)
    Kmm = _sprites_kernel64(z, z, None, False, se, True, La)
    Sr, vr = _stats64(Knm, means, vars_)
    inv_r = np.linalg.inv(Kmm[None] + Sr)
    mt_r = np.einsum("lij,lj->li", inv_r, vr)
    # the predictive quantities the caller forms (:1165, :624-628): K_bm mean_term and diag(K_bm Sigma^-1 K_mb)
    kb = Knm[:64]
    assert _rel(kb @ mt.cpu().numpy().astype(np.float64).T, kb @ mt_r.T) < 2e-3
    q = np.einsum("bi,lij,bj->lb", kb, inv.cpu().numpy().astype(np.float64), kb)
    assert _rel(q, np.einsum("bi,lij,bj->lb", kb, inv_r, kb)) < 2e-3
