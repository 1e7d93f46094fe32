"""Data-parallel epoch loop of the CLI drivers (svgp_vae_amd.dp.run_sharded_epochs) with real processes:
world-size 2 and 3 `gloo` ranks walk two epochs of un-shuffled batches whose last one is ragged (16 + 16 + 9 rows), every rank
on its contiguous row range of every batch, gradients through engine.DataParallelStep over the oracle-backed phase backend of
tests/test_dp_gloo.py, TF1 Adam and the GECO state machine of MNIST_experiment.py:313-355 on every rank -- and must reproduce
oracle.train_trajectory of ONE process on the same batches: the per-step elbo / recon_loss / C_ma / lagrange_mult log and the
parameters after the 6 updates.  What this pins: the row sharding incl. the ragged batch, c = N_train / b_global per batch,
that the scalars a rank logs are the global ones, that only rank 0 prints, and that ranks stay in step (no rank skips a
collective)."""
import math
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import svgpvae_oracle as O
from tests import helpers as H
from tests.test_dp_gloo import ORDER, OracleBackend, _free_port

DT = torch.float64
N_ROWS, BATCH, L, LR, N_TRAIN = 41, 16, 3, 0.004, 41.0


def _problem():
    params, images, aux, _ = H.toy_problem(b=N_ROWS, m=12, L=L, M=4, n_obj=20, seed=2)
    spans = [(lo, min(lo + BATCH, N_ROWS)) for lo in range(0, N_ROWS, BATCH)]
    eps_of = lambda epoch, i, b: torch.randn(b, L, dtype=DT, generator=torch.Generator().manual_seed(100 * epoch + i))
    return params, images, aux, spans, eps_of


def _worker(rank, world, port, geco, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from svgp_vae_amd.dp import DistContext, run_sharded_epochs
    from svgp_vae_amd.engine import DataParallelStep
    ctx = DistContext(backend="gloo").init()
    try:
        assert (ctx.rank, ctx.world, ctx.multi) == (rank, world, True) and dist.get_backend() == "gloo"
        params, images, aux, spans, eps_of = _problem()
        params = {k: v.clone() for k, v in params.items()}
        ms = {k: torch.zeros_like(v) for k, v in params.items()}
        vs = {k: torch.zeros_like(v) for k, v in params.items()}
        st = dict(C_ma=torch.zeros((), dtype=DT), lagr=torch.ones((), dtype=DT), first=True, t=0)
        said = []

        def local_step(llo, lhi, lo, hi, epoch, i):
            eps = eps_of(epoch, i, hi - lo)
            alpha = 0.0 if (geco and st["first"]) else 0.99
            be = OracleBackend(params, images[llo:lhi], aux[llo:lhi], eps[llo - lo:lhi - lo], b_global=hi - lo, rank=rank,
                               N_train=N_TRAIN, jitter=1e-6, geco=geco, beta=0.001, lagrange=float(st["lagr"]))
            DataParallelStep(be).step()                     # three gloo all-reduces: statA, statB, gradC
            flat, grads, off = be.block("gradC"), {}, 0
            for k in ORDER:
                n = params[k].numel()
                grads[k] = flat[off:off + n].view(params[k].shape)
                off += n
            assert float(flat[off + 3]) == hi - lo          # the all-reduced row count is the GLOBAL batch
            # scalars + GECO state of the step from the oracle on the global batch (test infrastructure: the exchange carries the
            # sums they are made of; their assembly is the HIP epilogue's job and is tested on the GPU)
            out, gfull = O.loss_and_grads(params, images[lo:hi], aux[lo:hi], eps, beta=0.001, C_ma=st["C_ma"], lagrange_mult=st["lagr"],
                                          alpha=alpha, kappa=math.sqrt(0.02), clipping_qs=True, GECO=geco, jitter=1e-6,
                                          N_train=N_TRAIN, L=L, formulation="efficient")
            for k in ORDER:                                  # the sharded exchange produced the global gradient
                assert H.relerr(grads[k], gfull[k]) < 1e-7, k
            st["t"] += 1
            O.adam_tf1_step(params, grads, ms, vs, st["t"], LR)
            if geco:
                st["C_ma"], st["lagr"] = out[13], out[14]
            st["first"] = False
            return dict(elbo=float(out[0]), recon_loss=float(out[1]), c_ma=float(out[13]), lagrange=float(out[14]))

        log = run_sharded_epochs(ctx, spans, 2, local_step, N_train=N_TRAIN, say=said.append)
        assert (len(said) == 2) == (rank == 0)               # rank 0 alone prints
        assert [s["rows"] for s in log["steps"]] == [16, 16, 9] * 2
        from svgp_vae_amd.engine import shard_rows
        assert [s["local_rows"] for s in log["steps"]] == [shard_rows(n, world, rank)[1] - shard_rows(n, world, rank)[0] for n in (16, 16, 9)] * 2
        ret.put((rank, log["steps"], {k: v.clone().numpy() for k, v in params.items()}))
    finally:
        ctx.close()


@pytest.mark.parametrize("world,geco", [(2, True), (2, False), (3, True)])
def test_sharded_epoch_loop_reproduces_the_single_process_trajectory(world, geco):
    mpc = mp.get_context("spawn")
    ret = mpc.SimpleQueue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, geco, ret)) for r in range(world)]
    for p in procs:
        p.start()
    import time
    got, t0 = [], time.time()
    while len(got) < world:
        if not ret.empty():
            got.append(ret.get())
            continue
        if any(p.exitcode not in (None, 0) for p in procs) or time.time() - t0 > 600:
            for q in procs:
                if q.is_alive():
                    q.terminate()
            pytest.fail(f"worker exit codes {[p.exitcode for p in procs]}")
        time.sleep(0.05)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    params, images, aux, spans, eps_of = _problem()
    batches = [(images[lo:hi], aux[lo:hi]) for _ in range(2) for lo, hi in spans]
    epsilons = [eps_of(e, i, hi - lo) for e in range(2) for i, (lo, hi) in enumerate(spans)]
    olog, oparams, _, _ = O.train_trajectory(params, batches, epsilons, beta=0.001, lr=LR, alpha_flag=0.99, kappa=math.sqrt(0.02),
                                             clipping_qs=True, GECO=geco, jitter=1e-6, N_train=N_TRAIN, L=L, formulation="efficient")
    for rank, steps, pfinal in got:
        for t, (g, w) in enumerate(zip(steps, olog)):
            for k in ("elbo", "recon_loss", "C_ma", "lagrange_mult"):
                assert abs(g[k] - w[k]) <= 1e-8 * max(1.0, abs(w[k])), (rank, t, k, g[k], w[k])
        for k, v in oparams.items():
            assert H.relerr(torch.from_numpy(pfinal[k]), v) < 1e-8, (rank, k)


def test_shard_batch_keeps_groups_whole_and_covers_the_batch():
    from svgp_vae_amd.dp import shard_batch
    for lo, hi, world, grp in ((0, 500, 8, 50), (500, 900, 3, 50), (512, 722, 8, 1), (0, 7, 8, 1)):
        spans = [shard_batch(lo, hi, world, r, grp) for r in range(world)]
        assert spans[0][0] == lo and spans[-1][1] == hi
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert all((b - a) % grp == 0 for a, b in spans)
        sizes = [(b - a) // grp for a, b in spans]
        assert max(sizes) - min(sizes) <= 1 and sorted(sizes, reverse=True) == sizes      # remainder to the lowest ranks
    with pytest.raises(AssertionError):
        shard_batch(0, 120, 2, 0, 50)
