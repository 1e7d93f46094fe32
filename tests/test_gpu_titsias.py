"""SVGPVAE_Titsias (mainSVGP(titsias=True), SVGPVAE_model.py:246-259 and :882-883) on the GPU against the oracle's
LITERAL restatement (b x b covariance, its inverse and Cholesky, per channel).  The HIP path never forms a b x b
matrix (Woodbury in m x m space, gp_titsias.hip); the 16-tuple scalars, p_m / p_v / recon and every parameter
gradient must agree, for GECO and beta-ELBO, on the LDS path (m <= 64) and the global-memory path (m = 72), and under
row sharding (virtual ranks, as tests/test_gpu_dp_virtual.py)."""
import math

import pytest
import torch

from oracle import svgpvae_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
DT = torch.float64


def _oracle(p, geco, jitter, N_train, C_ma=0.02, lagrange=1.3, alpha=0.9, beta=0.001, kappa2=0.02):
    params, images, aux, eps = p
    return O.loss_and_grads(params, images, aux, eps, beta=beta, C_ma=torch.tensor(C_ma, dtype=DT),
                            lagrange_mult=torch.tensor(lagrange, dtype=DT), alpha=alpha, kappa=math.sqrt(kappa2),
                            clipping_qs=True, GECO=geco, jitter=jitter, N_train=N_train, L=eps.shape[1], titsias=True)


@pytest.mark.parametrize("geco,b,m,L,M,jitter", [(True, 40, 12, 3, 4, 1e-4), (False, 40, 12, 3, 4, 1e-4),
                                                (True, 100, 32, 4, 8, 1e-3), (False, 90, 72, 2, 16, 1e-3)])
def test_titsias_step_matches_literal_oracle(geco, b, m, L, M, jitter):
    p = H.toy_problem(b=b, m=m, L=L, M=M, n_obj=20, seed=m + L)
    params, images, aux, eps = p
    N_train = 500.0
    out, grads = _oracle(p, geco, jitter, N_train)
    eng = H.engine_for(params, b, geco=geco, clip_qs=True, N_train=N_train, jitter=jitter, titsias=True)
    eng.set_scalars(c_ma=0.02, lagrange=1.3, alpha=0.9)
    dev = eng.device
    eng.bind(images.to(dev), aux.to(dev), eps.to(dev))
    eng.run(adam=False)
    eng.synchronize()
    sc = eng.scalars()
    bad = []
    for key, idx in (("elbo", 0), ("recon_loss", 1), ("kl_term", 2), ("inside_elbo", 3), ("ce_term", 4),
                     ("inside_recon", 10), ("inside_kl", 11)):
        want = float(out[idx])
        if not abs(sc[key] - want) <= 1e-8 * max(1.0, abs(want)):
            bad.append(f"scalar {key}: got {sc[key]!r} want {want!r}")
    for name, idx, shp in (("p_m", 5, (b, L)), ("p_v", 6, (b, L)), ("z", 12, (b, L)), ("recon", 9, (b, 784))):
        e = H.relerr(eng.ws_view(name, shp), out[idx].reshape(shp))
        if not e < 1e-8:
            bad.append(f"{name}: rel {e:.3e}")
    g = eng.grads()
    for k, want in grads.items():
        e = H.relerr(g[k], want)
        if not e < 1e-6:
            bad.append(f"grad {k}: rel {e:.3e} (max|want| {float(want.abs().max()):.3e})")
    assert not bad, "\n".join(bad)


def test_titsias_variational_loss_method_matches_reference_surface():
    """mnistSVGP(titsias=True).variational_loss -> (L_2, 0) for one channel."""
    from svgp_vae_amd.SVGPVAE_model import mnistSVGP
    params, images, aux, eps = H.toy_problem(b=50, m=16, L=2, M=4, n_obj=20, seed=3)
    jitter, N_train = 1e-4, 500.0
    svgp = mnistSVGP(True, False, params["inducing_index_points"].numpy(), False, params["object_vectors"].numpy(),
                     'main', jitter, N_train, 2)
    svgp.l_GP, svgp.amplitude = params["l_GP"].clone(), params["amplitude"].clone()
    y = torch.randn(50, dtype=DT, generator=torch.Generator().manual_seed(1))
    noise = torch.rand(50, dtype=DT, generator=torch.Generator().manual_seed(2)) + 0.05
    osvgp = O.MnistSVGP(True, params["inducing_index_points"], params["object_vectors"], params["l_GP"],
                        params["amplitude"], jitter, N_train)
    want, zero = osvgp.variational_loss(aux, y, None, None, noise)
    got, gz = svgp.variational_loss(aux, y, None, None, noise=noise)
    assert abs(float(got) - float(want)) <= 1e-9 * abs(float(want)) and float(gz) == 0.0 == float(zero)


@pytest.mark.parametrize("G", [2, 3])
def test_titsias_virtual_ranks_equal_single_engine(G):
    from svgp_vae_amd.engine import shard_rows
    from tests.test_gpu_dp_virtual import _lockstep
    b = 96
    params, images, aux, eps = H.toy_problem(b=b, m=24, L=3, M=4, n_obj=20, seed=7)
    kw = dict(geco=True, clip_qs=True, N_train=500.0, jitter=1e-4, titsias=True)
    single = H.engine_for(params, b, **kw)
    dev = single.device
    di, da, de = images.to(dev), aux.to(dev), eps.to(dev)
    single.bind(di, da, de)
    ranks = []
    for r in range(G):
        lo, hi = shard_rows(b, G, r)
        e = H.engine_for(params, hi - lo, rank=r, world_size=G, **kw)
        e.set_batch_size(hi - lo, b)
        e.bind(di[lo:hi].contiguous(), da[lo:hi].contiguous(), de[lo:hi].contiguous())
        ranks.append(e)
    for step in range(2):
        single.run(adam=True)
        single.synchronize()
        _lockstep(ranks, adam=True)
        ref = single.scalars()
        for e in ranks:
            sc = e.scalars()
            for k in ("elbo", "kl_term", "inside_elbo", "ce_term", "c_ma", "lagrange"):
                assert abs(sc[k] - ref[k]) <= 1e-9 * max(1.0, abs(ref[k])), (step, k, sc[k], ref[k])
            assert H.relerr(e.theta, single.theta) < 1e-7      # Adam divides by sqrt(v): reduction-order noise of tiny gradients is amplified
