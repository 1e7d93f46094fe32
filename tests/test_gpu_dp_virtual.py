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
