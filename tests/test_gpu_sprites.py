"""SPRITES SVGPVAE_Hensman step (spritesVAE + representation network + spritesSVGP product kernel) on the GPU
against the oracle restatement: the 16-tuple, and every gradient (networks, inducing points, GPLVM action table,
SE hyper-parameters), for the linear / cosine-normalised / squared-exponential kernels, both GP paths (m <= 64 LDS,
m = 72 global-memory GEMM), beta-ELBO and GECO, with the reference's p_v clipping and gradient clipping."""
import math

import numpy as np
import pytest
import torch

from oracle import sprites_oracle as SO

pytestmark = pytest.mark.gpu
DT = torch.float64


def _problem(b, frames, L, La, Lc, m, n_act, seed):
    g = torch.Generator().manual_seed(seed)
    params = {k: torch.tensor(v, dtype=DT) for k, v in SO.glorot_init(L, Lc, seed).items()}
    for k in params:
        if k.endswith("_b"):
            params[k] = 0.05 * torch.randn(*params[k].shape, dtype=DT, generator=g)
    gp = dict(inducing_index_points=torch.randn(m, La + Lc, dtype=DT, generator=g) * 1.5,
              GPLVM_action=torch.randn(n_act, La, dtype=DT, generator=g) * 1.5,
              # SE scales chosen so that kernel values are O(0.1..1) for N(0,1.5^2) features (distances ~6-8)
              l_action=torch.tensor(5.0, dtype=DT), sigma_action=torch.tensor(1.4, dtype=DT),
              l_character=torch.tensor(7.0, dtype=DT), sigma_character=torch.tensor(1.2, dtype=DT))
    images = torch.rand(b, 64, 64, 3, dtype=DT, generator=g)
    ids = torch.randint(0, n_act, (b,), generator=g)
    eps = torch.randn(b, L, dtype=DT, generator=g)
    seg, rep = SO.aux_data_sprites_utils(b, frames, frames)
    return params, gp, images, ids, eps, seg, rep


def _rel(a, c):
    a, c = torch.as_tensor(a, dtype=DT).cpu().reshape(-1), torch.as_tensor(c, dtype=DT).cpu().reshape(-1)
    return float((a - c).abs().max() / max(float(c.abs().max()), 1e-9))     # gradients below 1e-9 are noise


@pytest.mark.parametrize("K_SE,norm,GECO,m,clip,titsias", [
    (False, False, False, 10, None, False), (False, True, True, 12, None, False), (True, False, True, 10, 0.05, False),
    (False, False, True, 72, None, False),
    (True, False, True, 10, None, True), (False, True, False, 72, None, True)])      # SVGPVAE_Titsias (:246-259)
def test_sprites_step_matches_oracle(K_SE, norm, GECO, m, clip, titsias):
    from svgp_vae_amd import sprites as S
    b, frames, L, La, Lc, n_act = 8, 4, 6, 8, 16, 9
    params, gp, images, ids, eps, seg, rep = _problem(b, frames, L, La, Lc, m, n_act, seed=m + int(K_SE))
    kappa, jitter, N_train = math.sqrt(0.0075), 0.01, 100.0
    kw = dict(beta=0.001, C_ma=torch.tensor(0.02, dtype=DT), lagrange_mult=torch.tensor(1.4, dtype=DT), alpha=0.9,
              kappa=kappa, L=L, L_action=La, jitter=jitter, N_train=N_train, segment_ids=seg, repeats=rep,
              clipping_qs=True, GECO=GECO, K_obj_normalize=norm, K_SE=K_SE, clip_grad=clip, titsias=titsias)
    want, wgrads = SO.loss_and_grads(params, gp, (images, ids), eps, formulation="efficient", **kw)

    vae = S.spritesVAE(L)
    rnn = S.sprites_representation_network(Lc)
    svgp = S.spritesSVGP(titsias, False, gp["inducing_index_points"].numpy(), 'main', jitter, N_train, La,
                         gp["GPLVM_action"].numpy(), Lc, L, fixed_GP_params=False, fixed_GPLVM=False,
                         K_obj_normalize=norm, K_SE=K_SE)
    init = dict(params)
    init["se"] = torch.stack([gp["l_action"], gp["sigma_action"], gp["l_character"], gp["sigma_character"]])
    eng = S.SpritesStepEngine(vae, rnn, svgp, b_max=b, seg_len=frames, clip_qs=True, geco=GECO,
                              kappa_squared=0.0075, beta=0.001, clip_grad=clip, params=init)
    eng.set_scalars(c_ma=0.02, lagrange=1.4, alpha=0.9)
    dev = eng.dev
    eng.step(images.to(dev), ids.to(dev, DT), eps.to(dev), adam=False)
    got = eng.outputs()
    bad = []
    for i in range(16):
        e = _rel(got[i], want[i])
        if not e < 1e-8:
            bad.append(f"tuple[{i}] rel {e:.3e}")
    g = eng.grads
    for k, w in wgrads.items():
        if k in ("l_action", "sigma_action", "l_character", "sigma_character"):
            continue
        e = _rel(g[k], w)
        if not e < 1e-6:
            bad.append(f"grad {k} rel {e:.3e} (max {float(w.abs().max()):.2e})")
    if K_SE:
        wse = torch.stack([wgrads[k] for k in ("l_action", "sigma_action", "l_character", "sigma_character")])
        e = _rel(g["se"], wse)
        if not e < 1e-6:
            bad.append(f"grad se rel {e:.3e}")
    assert not bad, "\n".join(bad)


def test_sprites_training_steps_reduce_the_loss():
    from svgp_vae_amd import sprites as S
    b, frames, L, La, Lc, m, n_act = 8, 4, 6, 8, 16, 18, 9
    params, gp, images, ids, eps, seg, rep = _problem(b, frames, L, La, Lc, m, n_act, seed=3)
    svgp = S.spritesSVGP(False, False, gp["inducing_index_points"].numpy(), 'main', 0.01, 100.0, La,
                         gp["GPLVM_action"].numpy(), Lc, L, K_obj_normalize=True)
    eng = S.SpritesStepEngine(S.spritesVAE(L), S.sprites_representation_network(Lc), svgp, b_max=b, seg_len=frames,
                              geco=True, lr=3e-3, clip_grad=1e6)
    dev = eng.dev
    di, da = images.to(dev), ids.to(dev, DT)
    losses = []
    for _ in range(8):
        eng.step(di, da, None, adam=True)
        losses.append(eng.scalars()["recon_loss"])
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert eng.scalars()["adam_t"] == 8.0


@pytest.mark.parametrize("net_dtype", [torch.float64, torch.float32])
def test_sprites_side_streams_equal_one_stream(net_dtype):
    """m > 64: three Adam steps with the side branches (forward-factor tail + early reverse half, and the encoder reverse pass
    beside the kernel-matrix / representation-network reverse pass) against the same engine with every launch on one stream:
    identical parameters and Adam state, bit for bit (the same kernels on the same values, only the order of issue differs)."""
    from svgp_vae_amd import sprites as S
    b, frames, L, La, Lc, m, n_act = 24, 4, 6, 8, 16, 72, 9
    params, gp, images, ids, eps, seg, rep = _problem(b, frames, L, La, Lc, m, n_act, seed=5)
    out = []
    for one_stream in (False, True):
        svgp = S.spritesSVGP(False, False, gp["inducing_index_points"].numpy(), 'main', 0.01, 100.0, La,
                             gp["GPLVM_action"].numpy(), Lc, L, K_obj_normalize=True)
        eng = S.SpritesStepEngine(S.spritesVAE(L), S.sprites_representation_network(Lc), svgp, b_max=b, seg_len=frames,
                                  geco=True, lr=3e-3, clip_grad=1e6, net_dtype=net_dtype)
        assert eng.side is not None and eng.scratch2 is not None
        if one_stream:
            eng.side = eng.side2 = None
        dev = eng.dev
        di, da, de = images.to(dev, net_dtype), ids.to(dev, DT), eps.to(dev)
        for _ in range(3):
            eng.step(di, da, de, adam=True)
        eng.stream.synchronize()
        out.append((eng.theta.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.state.clone()))
    for x, y in zip(*out):
        assert torch.equal(x, y)


@pytest.mark.parametrize("G,K_SE,m,clip,shard,L,pack", [(2, False, 12, None, None, 6, "0"), (3, True, 72, 0.05, None, 6, "0"),
                                                         (2, False, 72, None, False, 6, "0"), (2, True, 72, None, True, 6, "0"),
                                                         (8, False, 72, None, True, 8, "0"), (3, True, 72, 0.05, None, 6, "1"),
                                                         (8, False, 72, None, True, 8, "1")])
def test_sprites_virtual_ranks_equal_single_engine(G, K_SE, m, clip, shard, L, pack, monkeypatch):
    """Data parallelism over whole character groups: G engines on one GPU run the step's phases in
    lockstep and every exchange point is executed by hand (engine.virtual_exchange: what the RCCL collectives do);
    scalars, gradients and the parameters after two Adam steps must equal the single-engine run at the same global
    batch.  m <= 64: three all-reduces.  m > 64 and L divisible by G (shard None -> on): the channel-sharded schedule --
    reduce-scatter of S, v / A2, ud, td over the channels, each rank factors its L / G channels
    (svgp_gp_factor_*_channels), all-gather of Sigma^-1, M2, t, u / Ssym, vbar, KL; pack = 1: the symmetric (L,m,m) members
    tile-packed (SVGP_DP_PACK), the window's tail and early reverse half on the side stream beside the exchange."""
    from svgp_vae_amd import sprites as S
    monkeypatch.setenv("SVGP_DP_PACK", pack)
    frames, La, Lc, n_act = 4, 8, 16, 9
    b = frames * 2 * G
    params, gp, images, ids, eps, _, _ = _problem(b, frames, L, La, Lc, m, n_act, seed=G + m)
    init = dict(params)
    init["se"] = torch.stack([gp["l_action"], gp["sigma_action"], gp["l_character"], gp["sigma_character"]])

    def make(rank, world, b_max):
        svgp = S.spritesSVGP(False, False, gp["inducing_index_points"].numpy(), 'main', 0.01, 100.0, La,
                             gp["GPLVM_action"].numpy(), Lc, L, fixed_GP_params=False, fixed_GPLVM=False,
                             K_obj_normalize=not K_SE, K_SE=K_SE)
        e = S.SpritesStepEngine(S.spritesVAE(L), S.sprites_representation_network(Lc), svgp, b_max=b_max,
                                seg_len=frames, clip_qs=True, geco=True, kappa_squared=0.0075, clip_grad=clip,
                                params=init, rank=rank, world_size=world, channel_shard=shard if world > 1 else False)
        e.set_scalars(c_ma=0.02, lagrange=1.4, alpha=0.9)
        return e

    single = make(0, 1, b)
    dev = single.dev
    di, da, de = images.to(dev), ids.to(dev, DT), eps.to(dev)
    ranks = [make(r, G, b // G) for r in range(G)]
    assert ranks[0].chan_shard == (m > 64 and shard is not False)
    rows = [slice(r * (b // G), (r + 1) * (b // G)) for r in range(G)]
    for step in range(2):
        single.step(di, da, de, adam=True)
        gens = [e.phases(di[sl].contiguous(), da[sl].contiguous(), de[sl].contiguous(), True, b)
                for e, sl in zip(ranks, rows)]
        from svgp_vae_amd.engine import virtual_exchange
        while True:
            ops = [next(gn, None) for gn in gens]
            if ops[0] is None:
                assert all(o is None for o in ops)
                break
            for e in ranks:
                e.stream.synchronize()
            virtual_exchange(ops)
            torch.cuda.synchronize()
        ref = single.scalars()
        for e in ranks:
            sc = e.scalars()
            for k in ("elbo", "recon_loss", "kl_term", "ce_term", "c_ma", "lagrange"):
                assert abs(sc[k] - ref[k]) <= 1e-9 * max(1.0, abs(ref[k])), (step, k, sc[k], ref[k])
            assert _rel(e.grad, single.grad) < 1e-8
            assert _rel(e.theta, single.theta) < 1e-9
