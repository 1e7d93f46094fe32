"""Moving-ball SVGP (SVGPVAE_model.py:17-171, 638-715): known-answer tests of the restatement.
The identities below are what lets the HIP path reuse the channel-batched GP stages for the ball: the T frames of
a video are the rows, the videos of the batch are the channels, N_train = T (c = 1), the kernel matrices are shared
because every video has the same time stamps 1..tmax (SVGPVAE_model.py:663-664)."""
import math

import pytest
import torch

from oracle import ball_oracle as B
from oracle import staged_gp as SG
from oracle import svgpvae_oracle as O

DT = torch.float64


def _toy(batch=5, T=12, m=6, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.arange(T, dtype=DT) + 1.0).repeat(batch, 1)
    y = torch.randn(batch, T, dtype=DT, generator=g)
    noise = torch.rand(batch, T, dtype=DT, generator=g) * 2 + 0.05
    z = torch.linspace(1.0, float(T), m, dtype=DT) + 0.1 * torch.randn(m, dtype=DT, generator=g)
    return x, y, noise, z, torch.tensor(1.7, dtype=DT)


def _shared_kernel_matrices(x, z, ls):
    K = B.se_matrix(z[:, None], z[:, None], ls)
    Kn = B.se_matrix(x[0][:, None], z[:, None], ls)
    return K, Kn, torch.ones(x.shape[1], dtype=DT)


@pytest.mark.parametrize("jitter", [1e-9, 1e-4])
def test_ball_hensman_literal_equals_channel_batched_block(jitter):
    x, y, noise, z, ls = _toy()
    sv = B.BallSVGP(False, z, ls, jitter)
    mean, Bm, mu_hat, A_hat = sv.approximate_posterior_params(x, y, noise)
    L3, KL = sv.variational_loss(x, y, noise, mu_hat, A_hat)
    K, Kn, knn = _shared_kernel_matrices(x, z, ls)
    T = x.shape[1]
    p_m, p_v, L3e, KLe, aux = O.gp_block_efficient(K, Kn, knn, y.T.contiguous(), noise.T.contiguous(), jitter, float(T),
                                                  want_aux=True, kl_form=1)
    assert torch.allclose(mean, p_m.T, rtol=1e-7, atol=1e-9)
    assert torch.allclose(torch.diagonal(Bm, dim1=1, dim2=2), p_v.T, rtol=1e-7, atol=1e-9)
    assert torch.allclose(mu_hat, aux["mu_hat"], rtol=1e-7, atol=1e-9)
    assert torch.allclose(A_hat, aux["A_hat"], rtol=1e-7, atol=1e-9)
    assert torch.allclose(L3, L3e, rtol=1e-8)
    # the reference adds the batch-wide scalar to every video: same sum over videos, and the literal per-video value
    # is recovered from the per-channel pieces
    assert abs(float(KL.sum() - KLe.sum())) < 1e-8 * abs(float(KL.sum()))
    klq = torch.einsum('ij,ljk,lki->l', aux["Ki"], aux["A_hat"], aux["A_hat"])
    lit = KLe - 0.5 * x.shape[0] * klq + 0.5 * klq.sum()
    assert torch.allclose(KL, lit, rtol=1e-8)


@pytest.mark.parametrize("jitter", [1e-9, 1e-4])
def test_ball_titsias_literal_equals_woodbury(jitter):
    x, y, noise, z, ls = _toy(seed=1)
    sv = B.BallSVGP(True, z, ls, jitter)
    L2, zero = sv.variational_loss(x, y, noise, None, None)
    assert zero == 0.0
    K, Kn, knn = _shared_kernel_matrices(x, z, ls)
    L2e = O.titsias_block_efficient(K, Kn, knn, y.T.contiguous(), noise.T.contiguous(), jitter)
    assert torch.allclose(L2, L2e, rtol=1e-8)


def test_staged_backward_with_ball_kl_form_matches_autograd():
    x, y, noise, z, ls = _toy(batch=4, T=10, m=5, seed=2)
    K, Kn, knn = _shared_kernel_matrices(x, z, ls)
    g = torch.Generator().manual_seed(5)
    Tn, L = x.shape[1], x.shape[0]
    eps, zbar = torch.randn(Tn, L, dtype=DT, generator=g), torch.randn(Tn, L, dtype=DT, generator=g)
    N, j, gT = float(Tn), 1e-6, -0.41
    lv = [t.clone().requires_grad_() for t in (K, Kn, knn, y.T.contiguous(), noise.T.contiguous())]
    p_m, p_v, L3, KL = O.gp_block_efficient(*lv, j, N, kl_form=1)
    ce = O.gauss_cross_entropy(p_m, p_v, lv[3], lv[4]).sum()
    zz = p_m + eps * torch.sqrt(p_v)
    loss = gT * (-ce + L3.sum() - KL.sum()) + (zbar * zz).sum()
    gs = torch.autograd.grad(loss, lv)
    f, ps, fb, man = SG.gp_block_manual(K, Kn, knn, lv[3].detach(), lv[4].detach(), eps, zbar, gT, j, N, kl_form=1)
    assert torch.allclose(f["KL"], KL, rtol=1e-12)
    for a, b_ in zip(gs, man):
        assert float((a - b_).abs().max() / a.abs().max()) < 1e-9


def test_ball_elbo_assembly_and_gradients_run():
    """build_SVGPVAE_elbo_graph on a tiny batch: shapes of the returned tuple, elbo = recon + beta * KL_term, finite
    gradients for every trainable (MLPs, inducing points, length scales)."""
    from oracle import pearce_vae_oracle as P
    g = torch.Generator().manual_seed(0)
    batch, T, px, m = 3, 8, 8, 4
    vid = P.make_video_batch(tmax=T, px=px, py=px, batch=batch, r=2, generator=g, dtype=DT)
    p = {k: v.to(DT) for k, v in P.init_mlp_params(px, px, hidden=16, seed=0).items()}
    for c in "xy":
        p[f"ip_{c}"] = B.BallSVGP.initial_inducing_points(m, False, 1, T, 1, T)
        p[f"l_{c}"] = torch.tensor(2.0, dtype=DT)
    eps = torch.randn(batch, T, 2, dtype=DT, generator=g)
    for titsias in (False, True):
        out, loss, grads = B.loss_and_grads(p, vid, eps, beta=0.7, titsias=titsias, jitter=1e-6, clipping_qs=True)
        elbo, recon, KLt = out[0], out[1], out[2]
        assert elbo.shape == (batch,) and out[5].shape == (batch, T, 2) and out[9].shape == (batch, T, px, px)
        assert torch.allclose(elbo, recon + 0.7 * KLt)
        assert abs(float(loss + elbo.mean())) < 1e-12
        for k, v in grads.items():
            assert torch.isfinite(v).all(), k
            assert float(v.abs().max()) > 0, k
