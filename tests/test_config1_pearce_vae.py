"""BASELINE config 1 (`BALL_experiment.py --elbo VAE`, CPU plumbing): known-answer tests of
the Pearce GP-VAE restatement and a short CPU training run."""
import math

import torch

from oracle import pearce_vae_oracle as P
from oracle import svgpvae_oracle as O


def test_pearce_gp_degenerates_to_standard_normal_prior_at_tiny_length_scale():
    """lt = 0.001 (BALL_experiment.py:46-48): K = I + diag(var) => p_m = y/(1+var), p_v = var/(1+var),
    lhood = sum log N(y | 0, 1+var), and elbo_prior_kl = -KL(q || N(0,1))."""
    g = torch.Generator().manual_seed(0)
    b, t = 4, 30
    T = torch.arange(t, dtype=torch.float64).repeat(b, 1)
    y = torch.randn(b, t, dtype=torch.float64, generator=g)
    var = torch.rand(b, t, dtype=torch.float64, generator=g) + 0.1
    p_m, p_v, lh = P.build_1d_gp(T, y, var, T, 0.001)
    assert torch.allclose(p_m, y / (1 + var), atol=1e-12)
    assert torch.allclose(p_v, var / (1 + var), atol=1e-12)
    want = torch.distributions.Normal(0.0, (1 + var).sqrt()).log_prob(y).sum(1)
    assert torch.allclose(lh, want, atol=1e-10)
    ce = O.gauss_cross_entropy(p_m, p_v, y, var).sum(1)
    # q = the GP posterior N(p_m, p_v): log Z - E_q[log N(z | y, var)] = -KL(q || N(0, 1))
    kl = torch.distributions.kl_divergence(torch.distributions.Normal(p_m, p_v.sqrt()),
                                           torch.distributions.Normal(torch.zeros_like(y), torch.ones_like(y))).sum(1)
    assert torch.allclose(lh - ce, -kl, atol=1e-9)


def test_pearce_gp_matches_dense_gp_regression():
    g = torch.Generator().manual_seed(1)
    t = 12
    T = torch.arange(t, dtype=torch.float64)[None]
    y = torch.randn(1, t, dtype=torch.float64, generator=g)
    var = torch.rand(1, t, dtype=torch.float64, generator=g) + 0.05
    p_m, p_v, lh = P.build_1d_gp(T, y, var, T, 2.0)
    K = torch.exp(-0.5 * (T[0][:, None] - T[0][None, :]) ** 2 / 4.0)
    Kn = K + torch.diag(var[0])
    assert torch.allclose(p_m[0], K @ torch.linalg.solve(Kn, y[0]), atol=1e-10)
    assert torch.allclose(p_v[0], 1 - torch.diagonal(K @ torch.linalg.solve(Kn, K)), atol=1e-10)
    mvn = torch.distributions.MultivariateNormal(torch.zeros(t, dtype=torch.float64), Kn)
    assert abs(float(lh[0] - mvn.log_prob(y[0]))) < 1e-9


def test_video_batch_shapes_and_ball_size():
    vid = P.make_video_batch(generator=torch.Generator().manual_seed(2))
    assert vid.shape == (35, 30, 32, 32) and set(vid.unique().tolist()) <= {0.0, 1.0}
    assert 20 <= float(vid.sum((2, 3)).max()) <= 32           # an open radius-3 disc covers 25..32 pixels


def test_ball_vae_smoke_run_improves_the_elbo():
    """README's setup check `python BALL_experiment.py --elbo VAE`, 25 steps instead of 25 000."""
    elbos = P.run_ball_vae(steps=25, lr=1e-3, batch=12, tmax=30)
    assert all(math.isfinite(e) for e in elbos)
    assert sum(elbos[-5:]) / 5 > sum(elbos[:5]) / 5
