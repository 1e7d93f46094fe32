"""Data parallelism of the HIP step, checked on ONE GPU with virtual ranks: G engines (rank r of G, rows
shard_rows(b, G, r)) run the four phases in lockstep and the three exchange blocks are summed across them
by hand -- exactly what the RCCL all-reduce does between real ranks.  ELBO, every scalar of the epilogue and
the parameters after three Adam steps must match the single-engine run at the same global batch
(reduction-order tolerance only; SURVEY 8e "parity oracle for DP")."""
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _lockstep(engines, adam=True):
    for k in range(4):
        for e in engines:
            e.phase(k, adam)
        if k < 3:
            name = ("statA", "statB", "gradC")[k]
            for e in engines:
                e.synchronize()
            tot = sum(e.block(name) for e in engines)
            for e in engines:
                with torch.cuda.stream(e.stream):
                    e.block(name).copy_(tot)
                e.synchronize()


@pytest.mark.parametrize("G,b,geco", [(2, 256, True), (3, 210, True), (2, 256, False), (8, 256, True)])
def test_virtual_ranks_equal_single_engine(golden, G, b, geco):
    from svgp_vae_amd.engine import shard_rows
    params, images, aux, eps = H.golden_problem(golden, rows=slice(0, b))
    single = H.engine_for(params, b, geco=geco)
    dev = single.device
    di, da, de = images.to(dev), aux.to(dev), eps.to(dev)
    single.bind(di, da, de)
    ranks = []
    for r in range(G):
        lo, hi = shard_rows(b, G, r)
        e = H.engine_for(params, hi - lo, geco=geco, rank=r, world_size=G)
        e.set_batch_size(hi - lo, b)
        e.bind(di[lo:hi].contiguous(), da[lo:hi].contiguous(), de[lo:hi].contiguous())
        ranks.append(e)
    for step in range(3):
        single.run(adam=True)
        single.synchronize()
        _lockstep(ranks, adam=True)
        ref = single.scalars()
        for e in ranks:
            sc = e.scalars()
            for k in ("elbo", "recon_loss", "kl_term", "inside_elbo", "ce_term", "c_ma", "lagrange", "adam_t"):
                assert abs(sc[k] - ref[k]) <= 1e-9 * max(1.0, abs(ref[k])), (step, k, sc[k], ref[k])
            assert H.relerr(e.theta, single.theta) < 1e-9
    # replicas stay bit-identical to each other (same reduced inputs, same kernels)
    for e in ranks[1:]:
        assert torch.equal(e.theta, ranks[0].theta)


@pytest.mark.parametrize("G,b,m,L,M,pack", [(2, 64, 72, 4, 16, "0"), (4, 96, 72, 4, 16, "0"), (2, 64, 72, 4, 16, "1"),
                                            (3, 96, 130, 6, 24, "1"), (8, 1024, 256, 16, 32, None), (8, 1024, 256, 16, 32, "1")])
def test_channel_sharded_virtual_ranks_equal_single_engine(G, b, m, L, M, pack, monkeypatch):
    """Large-m path: the channel-sharded schedule of svgp_mnist_train_step_dp (reduce-scatter of the (L,m,m) statistics
    over the channels, every rank factors L / G channels, all-gather of what the row stages need), run stage
    by stage with G virtual ranks, against the single-engine step at the same global batch.  The last case is BASELINE
    configs[2] (m = 256, b = 1024, L = 16) on 8 ranks: 2 channels per rank.  pack: SVGP_DP_PACK -- the symmetric (L,m,m)
    members of the four large exchange points travel tile-packed (svgp_sym_pack; the window of Sigma^-1 / M2 goes through
    the HIP pack kernels inside the engine, M2 symmetrised by averaging); None = the library's default (on from m = 512).
    The forward tail and the early reverse half of every rank run on its side stream beside the exchange."""
    from svgp_vae_amd.engine import shard_rows, virtual_exchange, dp_pack_enabled
    if pack is None:
        monkeypatch.delenv("SVGP_DP_PACK", raising=False)
    else:
        monkeypatch.setenv("SVGP_DP_PACK", pack)
    assert dp_pack_enabled(m) == (pack == "1" or (pack is None and m >= 512))
    params, images, aux, eps = H.toy_problem(b=b, m=m, L=L, M=M, n_obj=60, seed=G + m)
    # The ranks add the statistics in a different order than the single engine and cond(Sigma_l) amplifies that rounding:
    # with c = N_train / b = 63, 1 / sigma^2 up to 1000 and jitter 1e-2, Sigma_l^-1 already differs by 1.4e-9 and the ELBO by
    # 9e-8 (measured).  c = 2 keeps the comparison about the schedule, not the conditioning.
    kw = dict(geco=True, N_train=2.0 * b, jitter=1e-2)
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
        assert e.channel_sharded()
        ranks.append(e)
    # noise floor: the same single-engine step on the rows in reversed order (mathematically identical; only the
    # summation order of the statistics changes -- which is also all that sharding changes)
    rev = H.engine_for(params, b, **kw)
    rev.bind(di.flip(0).contiguous(), da.flip(0).contiguous(), de.flip(0).contiguous())
    rev.run(adam=False); rev.synchronize()
    probe = H.engine_for(params, b, **kw)
    probe.bind(di, da, de)
    probe.run(adam=False); probe.synchronize()
    sr, sp = rev.scalars(), probe.scalars()
    noise = max(abs(sr[k] - sp[k]) / max(1.0, abs(sp[k])) for k in ("elbo", "kl_term", "inside_elbo"))
    gnoise = {k: H.relerr(rev.grads()[k], probe.grads()[k]) for k in probe.grads()}
    for step in range(2):
        single.run(adam=True)
        single.synchronize()
        gens = [e.sharded_stages(adam=True) for e in ranks]
        n_points = 0
        while True:
            ops = [next(g, None) for g in gens]
            if ops[0] is None:
                assert all(o is None for o in ops)
                break
            for e in ranks:
                e.synchronize()
            assert all(len(o) == len(ops[0]) for o in ops)
            if dp_pack_enabled(m) and n_points < 4:            # the (L,m,m) member of every large point is marked symmetric
                assert ops[0][0].sym is not None
                if m >= 256:                                   # 56 % of the square at m = 256 (52 % at m = 800)
                    assert ops[0][0].sym.xp.numel() < 0.6 * ops[0][0].tensor.numel()
            virtual_exchange(ops)
            torch.cuda.synchronize()
            n_points += 1
        assert n_points == 5
        for e in ranks:
            e.synchronize()
        ref = single.scalars()
        for e in ranks:
            sc = e.scalars()
            for k in ("elbo", "recon_loss", "kl_term", "inside_elbo", "ce_term", "c_ma", "lagrange", "adam_t"):
                assert abs(sc[k] - ref[k]) <= max(1e-8, 20 * noise) * max(1.0, abs(ref[k])), (step, k, sc[k], ref[k], noise)
            g, gs = e.grads(), single.grads()
            for k in gs:      # the GP parameters' gradients pass through (A_hat + jI)^-1: ill-conditioned (test_gpu_fullsize)
                tol = max(1e-5 if k in ("inducing_index_points", "l_GP", "amplitude", "object_vectors") else 1e-7, 20 * gnoise[k])
                assert H.relerr(g[k], gs[k]) < tol, (step, k, gnoise[k])
            # Adam's first updates are lr * g / (|g| + eps)-like: a gradient component of size 1e-10 with a 1e-8 relative
            # difference in the LARGE components next to it moves by a visible fraction of lr
            assert H.relerr(e.theta, single.theta) < max(1e-5, 1e3 * max(gnoise.values()))
    for e in ranks[1:]:
        assert torch.equal(e.theta, ranks[0].theta)


@pytest.mark.parametrize("G,b", [(2, 256), (8, 256), (3, 210)])
def test_split_gradient_exchange_virtual_ranks_equal_single_engine(golden, G, b):
    """cfg.split_grad_exchange: the closing all-reduce as two messages -- gradC[n_enc:] (decoder + GP parameters
    + scalar sums) after phase 4 (everything of phase 2 up to the kernel-matrix reverse pass + gradient reduction part 1),
    gradC[:n_enc] (encoder) after phase 5 (encoder reverse pass + reduction part 2) -- with G virtual ranks against the single
    engine's ordinary step: scalars and parameters after three Adam steps."""
    from svgp_vae_amd.engine import shard_rows
    params, images, aux, eps = H.golden_problem(golden, rows=slice(0, b))
    single = H.engine_for(params, b, geco=True)
    dev = single.device
    di, da, de = images.to(dev), aux.to(dev), eps.to(dev)
    single.bind(di, da, de)
    ranks = []
    for r in range(G):
        lo, hi = shard_rows(b, G, r)
        e = H.engine_for(params, hi - lo, geco=True, rank=r, world_size=G, split_grad_exchange=True)
        assert e.cfg.split_grad_exchange == 1
        e.set_batch_size(hi - lo, b)
        e.bind(di[lo:hi].contiguous(), da[lo:hi].contiguous(), de[lo:hi].contiguous())
        ranks.append(e)

    def exchange(name):
        for e in ranks:
            e.synchronize()
        tot = sum(e.block(name) for e in ranks)
        for e in ranks:
            with torch.cuda.stream(e.stream):
                e.block(name).copy_(tot)
            e.synchronize()

    for step in range(3):
        single.run(adam=True)
        single.synchronize()
        for k, name in ((0, "statA"), (1, "statB"), (4, "gradC_tail"), (5, "gradC_head"), (3, None)):
            for e in ranks:
                e.phase(k, True)
            if name:
                exchange(name)
        ref = single.scalars()
        for e in ranks:
            sc = e.scalars()
            for k in ("elbo", "recon_loss", "kl_term", "inside_elbo", "ce_term", "c_ma", "lagrange", "adam_t"):
                assert abs(sc[k] - ref[k]) <= 1e-9 * max(1.0, abs(ref[k])), (step, k, sc[k], ref[k])
            assert H.relerr(e.theta, single.theta) < 1e-9
    n_enc = int(ranks[0].pl.n_enc)
    assert ranks[0].block("gradC_head").numel() == n_enc
    assert ranks[0].block("gradC_head").numel() + ranks[0].block("gradC_tail").numel() == ranks[0].block("gradC").numel()
