"""In-library RCCL exchange (svgp_comm_*, svgp_mnist_train_step_dp) on one GPU: a 1-rank communicator
exercises the dlopen'd RCCL, the stream plumbing and the phase / all-reduce interleave; the step must be
bit-identical to svgp_mnist_train_step (a world-size-1 all-reduce is the identity).  The multi-rank
schedule itself is covered on CPU by tests/test_dp_gloo.py."""
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_unique_id_and_allreduce_identity():
    from svgp_vae_amd.engine import RcclComm
    uid = RcclComm.unique_id()
    assert len(uid) == 128
    comm = RcclComm(0, 1, uid)
    x = torch.arange(1000, dtype=torch.float64, device="cuda:0")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    comm.all_reduce(x, s.cuda_stream)
    s.synchronize()
    assert torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float64))
    comm.close()


def test_float32_allreduce_and_sharded_statistics():
    """svgp_allreduce_sum_f32 on a 1-rank communicator + the row-sharded form of the float32 statistics: two shards
    computed separately and added equal the unsharded pass (what the all-reduce does across ranks)."""
    from svgp_vae_amd import stream_stats as SS
    from svgp_vae_amd.engine import RcclComm
    comm = RcclComm(0, 1, RcclComm.unique_id())
    g = torch.Generator(device="cuda").manual_seed(0)
    n, m, L = 3000, 200, 3
    K = torch.randn(n, m, device="cuda", generator=g)
    mu = torch.randn(n, L, device="cuda", generator=g)
    var = torch.rand(n, L, device="cuda", generator=g) + 0.1
    S, v = SS.stats(K, mu, var, comm=comm)
    torch.cuda.synchronize()
    Sa, va = SS.stats(K[:1700].contiguous(), mu[:1700].contiguous(), var[:1700].contiguous())
    Sb, vb = SS.stats(K[1700:].contiguous(), mu[1700:].contiguous(), var[1700:].contiguous())
    assert float((Sa + Sb - S).abs().max() / S.abs().max()) < 1e-5
    assert float((va + vb - v).abs().max() / v.abs().max()) < 1e-5
    comm.close()


def test_dp_entry_equals_single_gpu_step(golden):
    from svgp_vae_amd.engine import RcclComm
    params, images, aux, eps = H.golden_problem(golden)
    a = H.engine_for(params, 256, geco=True)
    b_ = H.engine_for(params, 256, geco=True)
    b_.attach_comm(RcclComm(0, 1, RcclComm.unique_id()))
    dev = a.device
    di, da, de = images.to(dev), aux.to(dev), eps.to(dev)
    a.bind(di, da, de); b_.bind(di, da, de)
    for _ in range(3):
        a.run(adam=True)
        b_.run(adam=True)
    a.synchronize(); b_.synchronize()
    assert torch.equal(a.theta, b_.theta)
    assert a.scalars() == b_.scalars()
    b_.comm.close()
