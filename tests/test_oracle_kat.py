"""Known-answer tests that pin the CPU oracle without TensorFlow.

The reference has no tests; these are identities its own two branches / formulas must satisfy.
"""
import math

import numpy as np
import pytest
import torch

from oracle import staged_gp as SG
from oracle import svgpvae_oracle as O

DT = torch.float64


def _toy(b=40, m=12, L=3, M=4, n_obj=20, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, dtype=DT, generator=g)
    u = lambda *s: torch.rand(*s, dtype=DT, generator=g)
    ip = torch.cat([torch.arange(m, dtype=DT)[:, None], u(m, 1) * 6.28, r(m, M) * 1.5], 1)
    ov = r(n_obj, M) * 1.5
    ids = torch.randint(0, n_obj, (b, 1), generator=g).to(DT)
    aux = torch.cat([ids, u(b, 1) * 6.28, r(b, M)], 1)
    return dict(ip=ip, ov=ov, aux=aux, y=r(b, L), s2=u(b, L) * 2 + 0.05, eps=r(b, L), zbar=r(b, L))


def test_kernel_values():
    """ExpSinSquared(d=0)=a^2, (d=pi)=a^2 exp(-2/l^2); Linear = dot; normalised Linear diag = 1."""
    a, l = torch.tensor(0.7, dtype=DT), torch.tensor(1.3, dtype=DT)
    x = torch.tensor([0.3], dtype=DT)
    assert torch.allclose(O.exp_sin_squared(x, x, a, l), a ** 2 * torch.ones(1, 1, dtype=DT))
    k = O.exp_sin_squared(x, x + math.pi, a, l)
    assert torch.allclose(k, a ** 2 * torch.exp(-2 / l ** 2) * torch.ones(1, 1, dtype=DT))
    o = torch.randn(5, 4, dtype=DT)
    assert torch.allclose(O.linear_kernel(o, o), o @ o.T)
    assert torch.allclose(torch.diagonal(O.linear_kernel(o, o, normalize=True)), torch.ones(5, dtype=DT))
    assert torch.allclose(O.linear_kernel(o, o, normalize=True, diag_only=True), torch.ones(5, dtype=DT))
    # 2*pi periodicity of the view kernel
    assert torch.allclose(O.exp_sin_squared(x, x + 0.4, a, l), O.exp_sin_squared(x, x + 0.4 + 2 * math.pi, a, l))


def test_gauss_cross_entropy_and_kl_against_torch_distributions():
    g = torch.Generator().manual_seed(1)
    mu1, mu2 = torch.randn(7, 3, dtype=DT, generator=g), torch.randn(7, 3, dtype=DT, generator=g)
    v1, v2 = torch.rand(7, 3, dtype=DT, generator=g) + 0.1, torch.rand(7, 3, dtype=DT, generator=g) + 0.1
    q, p = torch.distributions.Normal(mu1, v1.sqrt()), torch.distributions.Normal(mu2, v2.sqrt())
    ce = -(torch.distributions.kl_divergence(q, p) + q.entropy())     # E_q[log p]
    assert torch.allclose(O.gauss_cross_entropy(mu1, v1, mu2, v2), ce, atol=1e-12)
    kl = torch.distributions.kl_divergence(q, torch.distributions.Normal(0 * mu1, 1 + 0 * v1)).sum()
    assert torch.allclose(O.KL_term_standard_normal_prior(mu1, v1), kl, atol=1e-12)


@pytest.mark.parametrize("jitter,tol", [(0.0, 1e-12), (1e-6, 1e-5)])
def test_hensman_with_optimal_q_equals_titsias(jitter, tol):
    """With b = N_train, L3 - KL at the amortised (mu_hat, A_hat) equals Titsias' L2
    (SVGPVAE_model.py:246-259 vs :261-301); exact at jitter 0."""
    t = _toy(b=60, m=10)
    N = 60.0
    one = torch.tensor(1.0, dtype=DT)
    H = O.MnistSVGP(False, t["ip"], t["ov"], one, one, jitter, N)
    T = O.MnistSVGP(True, t["ip"], t["ov"], one, one, jitter, N)
    y, s2 = t["y"][:, 0], t["s2"][:, 0]
    _, _, mu_hat, A_hat = H.approximate_posterior_params(t["aux"], t["aux"], y, s2)
    l3, kl = H.variational_loss(t["aux"], y, mu_hat, A_hat, s2)
    l2, _ = T.variational_loss(t["aux"], y, mu_hat, A_hat, s2)
    assert abs(float(l3 - kl - l2)) <= tol * abs(float(l2))


def test_literal_equals_efficient_gp_block():
    t = _toy()
    one = torch.tensor(1.1, dtype=DT)
    sv = O.MnistSVGP(False, t["ip"], t["ov"], one, 0.9 * one, 1e-6, 300.0)
    K, Kn, knn = SG.kernel_matrix_fwd(t["aux"], t["ip"], t["ov"], one, 0.9 * one)
    p_m, p_v, L3, KL = O.gp_block_efficient(K, Kn, knn, t["y"], t["s2"], 1e-6, 300.0)
    for l in range(t["y"].shape[1]):
        pm_l, pv_l, mu_hat, A_hat = sv.approximate_posterior_params(t["aux"], t["aux"], t["y"][:, l], t["s2"][:, l])
        l3, kl = sv.variational_loss(t["aux"], t["y"][:, l], mu_hat, A_hat, t["s2"][:, l])
        assert torch.allclose(pm_l, p_m[:, l], rtol=1e-9, atol=1e-11)
        assert torch.allclose(pv_l, p_v[:, l], rtol=1e-9, atol=1e-11)
        assert abs(float(l3 - L3[l])) < 1e-9 * abs(float(l3))
        assert abs(float(kl - KL[l])) < 1e-9 * abs(float(kl))


@pytest.mark.parametrize("jitter", [1e-6, 1e-3])
def test_titsias_literal_equals_woodbury(jitter):
    """Titsias L_2 (SVGPVAE_model.py:246-259): the literal b x b form (inverse + Cholesky of diag(var) + K_nm K_mm^-1
    K_mn + jI) equals the m x m Woodbury form the HIP path implements."""
    t = _toy(b=60, m=14)
    one = torch.tensor(1.1, dtype=DT)
    sv = O.MnistSVGP(True, t["ip"], t["ov"], one, 0.9 * one, jitter, 300.0)
    K, Kn, knn = SG.kernel_matrix_fwd(t["aux"], t["ip"], t["ov"], one, 0.9 * one)
    L2 = O.titsias_block_efficient(K, Kn, knn, t["y"], t["s2"], jitter)
    for l in range(t["y"].shape[1]):
        l2, zero = sv.variational_loss(t["aux"], t["y"][:, l], None, None, t["s2"][:, l])
        assert float(zero) == 0.0
        assert abs(float(l2 - L2[l])) < 1e-9 * abs(float(l2))


@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("use_ov", [False, True])
def test_staged_kernel_vjp_matches_autograd(normalize, use_ov):
    t = _toy()
    ls, amp = torch.tensor(1.3, dtype=DT), torch.tensor(0.8, dtype=DT)
    ov = t["ov"] if use_ov else None
    leaves = [t["ip"].clone().requires_grad_(), ls.clone().requires_grad_(), amp.clone().requires_grad_()]
    if use_ov:
        leaves.append(ov.clone().requires_grad_())
    sv = O.MnistSVGP(False, leaves[0], leaves[3] if use_ov else None, leaves[1], leaves[2], 1e-6, 1.0, normalize)
    K = sv.kernel_matrix(leaves[0], leaves[0])
    Kn = sv.kernel_matrix(t["aux"], leaves[0], x_inducing=False)
    knn = sv.kernel_matrix(t["aux"], t["aux"], False, False, diag_only=True)
    K2, Kn2, knn2 = SG.kernel_matrix_fwd(t["aux"], t["ip"], ov, ls, amp, normalize)
    for a, b_ in ((K, K2), (Kn, Kn2), (knn, knn2)):
        assert torch.allclose(a, b_, atol=1e-13)
    g = torch.Generator().manual_seed(3)
    gK, gKn, gknn = (torch.randn(*x.shape, dtype=DT, generator=g) for x in (K, Kn, knn))
    gs = torch.autograd.grad((gK * K).sum() + (gKn * Kn).sum() + (gknn * knn).sum(), leaves)
    d = SG.kernel_matrix_bwd(t["aux"], t["ip"], ov, ls, amp, gK, gKn, gknn, normalize)
    for a, b_ in zip(gs, d):
        assert torch.allclose(a, b_, rtol=1e-10, atol=1e-11)
    assert float(d[0][:, 0].abs().max()) == 0.0      # id column of the inducing points gets no gradient


def test_staged_gp_backward_matches_autograd():
    t = _toy()
    one = torch.tensor(1.0, dtype=DT)
    K, Kn, knn = SG.kernel_matrix_fwd(t["aux"], t["ip"], t["ov"], one, one)
    N, j, gT = 300.0, 1e-6, -0.37
    lv = [x.clone().requires_grad_() for x in (K, Kn, knn, t["y"], t["s2"])]
    p_m, p_v, L3, KL = O.gp_block_efficient(*lv, j, N)
    ce = O.gauss_cross_entropy(p_m, p_v, lv[3], lv[4]).sum()
    z = p_m + t["eps"] * torch.sqrt(p_v)
    loss = gT * (-ce + L3.sum() - (Kn.shape[0] / N) * KL.sum()) + (t["zbar"] * z).sum()
    gs = torch.autograd.grad(loss, lv)
    f, ps, fb, man = SG.gp_block_manual(K, Kn, knn, t["y"], t["s2"], t["eps"], t["zbar"], gT, j, N)
    assert torch.allclose(ps["p_m"], p_m, atol=1e-12) and torch.allclose(ps["p_v"], p_v, atol=1e-12)
    assert torch.allclose(ps["L3"], L3, rtol=1e-12) and torch.allclose(f["KL"], KL, rtol=1e-12)
    for a, b_ in zip(gs, man):
        assert float((a - b_).abs().max() / a.abs().max()) < 1e-10


def test_row_partition_plus_summed_statistics_equals_single_batch():
    """Data-parallel identity: k-way row partition + summed S, v (and backward A2, ud, td)
    reproduces the single-batch block."""
    t = _toy(b=48)
    one = torch.tensor(1.0, dtype=DT)
    K, Kn, knn = SG.kernel_matrix_fwd(t["aux"], t["ip"], t["ov"], one, one)
    N, j, gT, c = 300.0, 1e-6, -1.0, 300.0 / 48
    _, ps1, _, man1 = SG.gp_block_manual(K, Kn, knn, t["y"], t["s2"], t["eps"], t["zbar"], gT, j, N)
    parts = [slice(0, 16), slice(16, 40), slice(40, 48)]
    p = O.reciprocal_no_nan(t["s2"])
    S = sum(SG.gp_stats(Kn[s], p[s], (p * t["y"])[s])[0] for s in parts)
    v = sum(SG.gp_stats(Kn[s], p[s], (p * t["y"])[s])[1] for s in parts)
    f = SG.gp_factor_fwd(K, S, v, j, c)
    pss = [SG.gp_posterior_fwd(Kn[s], knn[s], t["y"][s], t["s2"][s], t["eps"][s], f, c) for s in parts]
    assert torch.allclose(torch.cat([q["p_m"] for q in pss]), ps1["p_m"], atol=1e-12)
    ws = [SG.gp_posterior_bwd_weights(t["y"][s], t["s2"][s], t["eps"][s], q, t["zbar"][s], gT, c)
          for s, q in zip(parts, pss)]
    st = [SG.gp_stats(Kn[s], w[0], w[2], c * w[1]) for s, w in zip(parts, ws)]
    A2, ud, td = (sum(x[i] for x in st) for i in range(3))
    fb = SG.gp_factor_bwd(K, S, v, f, A2, ud, td, gT, c, N, 48.0)
    rows = [SG.gp_posterior_bwd_rows(Kn[s], knn[s], t["y"][s], t["s2"][s], q, f, fb, w[0], w[1], w[2], gT, c)
            for s, q, w in zip(parts, pss, ws)]
    assert torch.allclose(fb["Kbar"], man1[0], rtol=1e-9, atol=1e-11)
    for i in range(4):
        assert torch.allclose(torch.cat([r[i] for r in rows]), man1[1 + i], rtol=1e-9, atol=1e-11)


def test_tf_same_padding_and_shapes():
    """Keras shapes of mnistVAE and SAME/VALID semantics."""
    w = {k: torch.tensor(v, dtype=DT) for k, v in O.glorot_uniform_init(16, 0).items()}
    assert sum(v.numel() for v in w.values()) == 5721
    vae = O.MnistVAE(w, 16)
    x = torch.randn(3, 28, 28, 1, dtype=DT)
    mu, var = vae.encode(x)
    assert mu.shape == (3, 16) and var.shape == (3, 16) and bool((var > 0).all())
    assert vae.decode(mu).shape == (3, 28, 28, 1)
    # stride-2 SAME on an even input pads only bottom/right
    xi = torch.randn(1, 4, 4, 1, dtype=DT)
    k = torch.randn(3, 3, 1, 1, dtype=DT)
    out = O._conv2d_nhwc(xi, k, None, 2, 'same')
    xp = torch.zeros(1, 5, 5, 1, dtype=DT); xp[:, :4, :4] = xi
    ref = O._conv2d_nhwc(xp, k, None, 2, 'valid')
    assert torch.allclose(out, ref)


def test_adam_tf1_formula():
    p = {"w": torch.tensor([1.0, -2.0], dtype=DT)}
    g = {"w": torch.tensor([0.5, 0.25], dtype=DT)}
    m = {"w": torch.zeros(2, dtype=DT)}; v = {"w": torch.zeros(2, dtype=DT)}
    O.adam_tf1_step(p, g, m, v, 1, 1e-3)
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    exp = torch.tensor([1.0, -2.0], dtype=DT) - lr_t * (0.1 * g["w"]) / (torch.sqrt(0.001 * g["w"] ** 2) + 1e-8)
    assert torch.allclose(p["w"], exp, atol=1e-15)


def test_oracle_reproduces_committed_golden(golden):
    """Guards the committed vectors against silent drift of the restatement."""
    gin, gout = golden
    params = {k[4:]: torch.tensor(v, dtype=DT) for k, v in gin.items() if k.startswith("vae_")}
    for k in ("inducing_index_points", "l_GP", "amplitude", "object_vectors"):
        params[k] = torch.tensor(gin[k], dtype=DT)
    img, aux, eps = (torch.tensor(gin[k][:256], dtype=DT) for k in ("images", "aux", "epsilon"))
    out, grads = O.loss_and_grads(params, img, aux, eps, beta=0.001, C_ma=torch.zeros((), dtype=DT),
                                  lagrange_mult=torch.ones((), dtype=DT), alpha=0.99, kappa=math.sqrt(0.02),
                                  clipping_qs=True, GECO=False, jitter=1e-6, N_train=4050.0, L=16,
                                  formulation="efficient")
    assert abs(float(out[0]) - float(gout["beta_elbo"])) < 1e-9 * abs(float(gout["beta_elbo"]))
    assert np.allclose(out[5].numpy(), gout["beta_p_m"], rtol=1e-8, atol=1e-10)
    assert np.allclose(grads["enc_c1_w"].numpy(), gout["beta_grad_enc_c1_w"], rtol=1e-6, atol=1e-10)
    assert np.allclose(grads["object_vectors"].numpy(), gout["beta_grad_object_vectors"], rtol=1e-6, atol=1e-10)


def test_staged_w_form_matches_autograd_and_the_lds_staging():
    """Round-4 large-m staging (oracle/staged_gp.py, "W form"): k^T Ki A Ki k evaluated as w^T Si w with w = K Ki k.  Forward
    values equal the efficient oracle's, every gradient equals autograd of it and the m <= 64 staging's hand-derived ones;
    the rank-local row sums make a k-way row partition reproduce the single batch with rep_weight 1 on one rank only."""
    t = _toy(b=48)
    one = torch.tensor(1.0, dtype=DT)
    K, Kn, knn = SG.kernel_matrix_fwd(t["aux"], t["ip"], t["ov"], one, one)
    N, j, gT = 300.0, 1e-6, -0.37
    lv = [x.clone().requires_grad_() for x in (K, Kn, knn, t["y"], t["s2"])]
    p_m, p_v, L3, KL = O.gp_block_efficient(*lv, j, N)
    ce = O.gauss_cross_entropy(p_m, p_v, lv[3], lv[4]).sum()
    z = p_m + t["eps"] * torch.sqrt(p_v)
    loss = gT * (-ce + L3.sum() - (Kn.shape[0] / N) * KL.sum()) + (t["zbar"] * z).sum()
    gs = torch.autograd.grad(loss, lv)
    f, ps, fb, man = SG.gp_block_manual_w(K, Kn, knn, t["y"], t["s2"], t["eps"], t["zbar"], gT, j, N)
    _, ps0, _, man0 = SG.gp_block_manual(K, Kn, knn, t["y"], t["s2"], t["eps"], t["zbar"], gT, j, N)
    assert torch.allclose(ps["p_m"], p_m, atol=1e-12) and torch.allclose(ps["p_v"], p_v, atol=1e-12)
    assert torch.allclose(ps["L3"], L3, rtol=1e-12) and torch.allclose(ps["d"], ps0["d"], rtol=1e-11, atol=1e-12)
    sym = lambda a: a + a.T if a.ndim == 2 and a.shape[0] == a.shape[1] else a      # K enters symmetrically: compare Kbar + Kbar^T
    for a, b_, c_ in zip(gs, man, man0):
        assert float((sym(a) - sym(b_)).abs().max() / sym(a).abs().max()) < 1e-10
        assert float((sym(c_) - sym(b_)).abs().max() / sym(c_).abs().max()) < 1e-10
    # ---- row partition: statistics summed, row-local sums kept per rank, replicated part counted once
    c = N / 48.0
    parts = [slice(0, 16), slice(16, 40), slice(40, 48)]
    p = O.reciprocal_no_nan(t["s2"])
    S = sum(SG.gp_stats(Kn[s], p[s], (p * t["y"])[s])[0] for s in parts)
    v = sum(SG.gp_stats(Kn[s], p[s], (p * t["y"])[s])[1] for s in parts)
    fw = SG.gp_factor_fwd(K, S, v, j, c)
    pss = [SG.gp_posterior_fwd_w(Kn[s], knn[s], t["y"][s], t["s2"][s], t["eps"][s], fw, c, K) for s in parts]
    ws = [SG.gp_posterior_bwd_weights(t["y"][s], t["s2"][s], t["eps"][s], q, t["zbar"][s], gT, c) for s, q in zip(parts, pss)]
    st = [SG.gp_stats(Kn[s], w[0], w[2], c * w[1]) for s, w in zip(parts, ws)]
    A2, ud, td = (sum(x[i] for x in st) for i in range(3))
    SW = SG.gp_sw_mspace(S, K, fw["Ki"])                       # from the summed statistic: no exchange of its own
    assert torch.allclose(SW, sum(SG.gp_sw_rows(q) for q in pss), rtol=1e-9, atol=1e-12)
    locs = [SG.gp_rows_local_w(Kn[s], q, w[0], gT, K, fw["Ki"]) for s, q, w in zip(parts, pss, ws)]
    fbs = [SG.gp_factor_bwd_w(K, v, fw, A2, SW, ud, td, loc, gT, c, N, 48.0, rep_weight=1.0 if r == 0 else 0.0)
           for r, loc in enumerate(locs)]
    Kbar = sum(x["Kbar"] for x in fbs)
    assert float((sym(Kbar) - sym(man[0])).abs().max() / sym(man[0]).abs().max()) < 1e-10
    rows = [SG.gp_posterior_bwd_rows_w(Kn[s], knn[s], t["y"][s], t["s2"][s], q, fw, fbs[0], loc, w[0], w[1], w[2], gT, c, K)
            for s, q, w, loc in zip(parts, pss, ws, locs)]
    for i in range(4):
        assert torch.allclose(torch.cat([r[i] for r in rows]), man[1 + i], rtol=1e-9, atol=1e-11)
    # ---- channel windows: every window's share with rep_weight 1, the local part on ONE window call only
    L = v.shape[0]
    loc_all = SG.gp_rows_local_w(Kn, ps, SG.gp_posterior_bwd_weights(t["y"], t["s2"], t["eps"], ps, t["zbar"], gT, N / 48.0)[0],
                                 gT, K, f["Ki"])
    zero = {k: torch.zeros_like(x) for k, x in loc_all.items()}
    gw = SG.gp_posterior_bwd_weights(t["y"], t["s2"], t["eps"], ps, t["zbar"], gT, c)
    B2f, udf, tdf = SG.gp_stats(Kn, gw[0], gw[2], c * gw[1])
    SWf = SG.gp_sw_rows(ps)
    Kb = 0
    for w0 in range(0, L, 2):
        sl = slice(w0, min(w0 + 2, L))
        fwin = {k: (x[sl] if (torch.is_tensor(x) and x.ndim >= 1 and x.shape[0] == L and k not in ("Ki",)) else x) for k, x in f.items()}
        Kb = Kb + SG.gp_factor_bwd_w(K, v[sl], fwin, B2f[sl], SWf[sl], udf[sl], tdf[sl], loc_all if w0 == 0 else zero, gT, c, N,
                                     48.0)["Kbar"]
    assert float((sym(Kb) - sym(man[0])).abs().max() / sym(man[0]).abs().max()) < 1e-10


def test_repr_nn_pretraining_oracle_first_step_is_a_sign_step_and_learns():
    """oracle.sprites_oracle.pretrain_repr_nn_trajectory (SPRITES_experiment.py:325-357): TF1 Adam's FIRST update is
    lr * sign(gradient) for every parameter (m / sqrt(v) = +-1 and lr_1 = lr sqrt(1 - b2) / (1 - b1) undo each other), checked on
    the dense bias whose gradient is softmax - one-hot averaged over the batch; a few epochs on separable characters reduce the
    loss."""
    import torch.nn.functional as F
    from oracle import sprites_oracle as SO
    DT = torch.float64
    g = torch.Generator().manual_seed(0)
    Lc, ncls, n = 16, 4, 16
    params = {k: torch.tensor(v, dtype=DT) for k, v in SO.glorot_init(4, Lc, 1).items()}
    chars = torch.arange(n) // 4
    base = torch.rand(4, 1, 8, 8, 3, dtype=DT, generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)
    frames = (0.8 * base + 0.2 * torch.rand(4, 4, 64, 64, 3, dtype=DT, generator=g)).reshape(n, 64, 64, 3)
    lr = 1e-2
    hist, p1, W1, b1, _, _ = SO.pretrain_repr_nn_trajectory(params, frames, chars, nr_epochs=1, lr=lr, batch_size=n, n_classes=ncls, seed=2)
    import math
    import numpy as np
    lim = math.sqrt(6.0 / (Lc + ncls))
    W0 = torch.tensor(np.random.RandomState(2).uniform(-lim, lim, (Lc, ncls)), dtype=DT)
    logits = SO.repr_nn(params, frames) @ W0
    gb = (F.softmax(logits, 1) - F.one_hot(chars, ncls).to(DT)).mean(0)
    assert torch.allclose(b1, -lr * torch.sign(gb), rtol=0, atol=1e-4 * lr)      # (eps = 1e-8 beside sqrt(v) ~ 1e-3)
    assert abs(hist[0][0] - float(F.cross_entropy(logits, chars))) < 1e-12
    hist2 = SO.pretrain_repr_nn_trajectory(params, frames, chars, nr_epochs=25, lr=lr, batch_size=8, n_classes=ncls, seed=2)[0]
    assert hist2[-1][0] < 0.7 * hist2[0][0] and hist2[-1][1] >= hist2[0][1]
