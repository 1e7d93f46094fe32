"""Data-parallel launch context of the *_experiment.py drivers and of bench.py (the reference is single-process:
MNIST_experiment.py:299,308 -- the sharded epoch loop is this build's).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port P \
        -m svgp_vae_amd.MNIST_experiment --elbo SVGPVAE_Hensman ...

One process per GPU.  Every batch of the reference's un-shuffled epoch loop (MNIST_experiment.py:313-355, utils.py:846-848,
incl. the ragged last one) is cut into G contiguous row ranges (engine.shard_rows: remainder rows to the lowest ranks; SPRITES:
whole 50-frame groups), c = N_train / b_global, parameters replicated, statistics and gradients all-reduced by the engines (RCCL
communicator of the library when it can be created on every rank, torch.distributed otherwise).  Rank 0 alone prints, evaluates
and writes test_metrics.txt / checkpoints; the other ranks wait at a barrier.  The launcher must start the ranks BEFORE any GPU
call (torch.distributed.run does); a process that has touched the GPU must never exec another program."""
import os

import torch


class DistContext:
    """rank / world size / local rank of this process as the launcher's environment states them (RANK, WORLD_SIZE, LOCAL_RANK),
    and the torch.distributed process group (created on first use, backend nccl = RCCL on a GPU box, gloo without one)."""

    def __init__(self, rank=None, world=None, local_rank=None, backend=None, force=None):
        env = os.environ
        self.rank = int(env.get("RANK", 0)) if rank is None else rank
        self.world = int(env.get("WORLD_SIZE", 1)) if world is None else world
        self.local_rank = int(env.get("LOCAL_RANK", self.rank)) if local_rank is None else local_rank
        self.backend = backend
        # force (SVGP_FORCE_DIST=1): take the multi-rank code path -- process group, communicator, in-stream collectives -- also
        # with ONE rank: the rehearsal of that path on a 1-GPU box (bench.py --force-dist is the same switch)
        force = (env.get("SVGP_FORCE_DIST", "0") == "1") if force is None else force
        self.multi = self.world > 1 or bool(force)
        self._owns_group = False

    @property
    def device(self):
        return f"cuda:{self.local_rank}" if torch.cuda.is_available() else "cpu"

    def init(self):
        """Creates the default process group when there are several ranks (idempotent)."""
        if not self.multi:
            return self
        import torch.distributed as dist
        if not dist.is_initialized():
            on_gpu = torch.cuda.is_available()
            if on_gpu:
                torch.cuda.set_device(self.local_rank)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            kw = dict(device_id=torch.device(self.device)) if on_gpu else {}
            dist.init_process_group(backend=self.backend or ("nccl" if on_gpu else "gloo"), rank=self.rank,
                                    world_size=self.world, **kw)
            self._owns_group = True
        return self

    def barrier(self):
        if self.multi:
            import torch.distributed as dist
            dist.barrier()

    def broadcast_object(self, obj, src=0):
        if not self.multi:
            return obj
        import torch.distributed as dist
        box = [obj if self.rank == src else None]
        dist.broadcast_object_list(box, src=src)
        return box[0]

    def close(self):
        if self._owns_group:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
            self._owns_group = False


def shard_batch(lo, hi, world, rank, group_rows=1):
    """Rows [lo', hi') of the batch [lo, hi) that rank `rank` of `world` takes: contiguous, remainder to the lowest ranks
    (engine.shard_rows); group_rows > 1 (SPRITES: 50 frames of one character share a GP group) keeps groups whole."""
    from .engine import shard_rows
    n = hi - lo
    assert n % group_rows == 0, f"batch of {n} rows is not a multiple of the group size {group_rows}"
    glo, ghi = shard_rows(n // group_rows, world, rank)
    return lo + glo * group_rows, lo + ghi * group_rows


def run_sharded_epochs(ctx, train_batches, nr_epochs, local_step, *, N_train, group_rows=1, on_epoch_end=None, say=print,
                       elbo_reduce="sum"):
    """The reference's epoch loop (MNIST_experiment.py:313-355: un-shuffled batches, ragged last batch, elbo / recon_loss /
    C_ma / lagrange_mult fetched every step) with every batch cut over the ranks of `ctx`.

    local_step(llo, lhi, lo, hi, epoch, i) runs ONE optimiser step on this rank's rows [llo, lhi) of the global batch
    [lo, hi) (c = N_train / (hi - lo); the exchanges are the engine's) and returns the step's scalars -- global values,
    identical on every rank: dict(elbo=, recon_loss=, c_ma=, lagrange=).  on_epoch_end(epoch, log) is called on EVERY rank
    (evaluation on rank 0 + barrier is the caller's choice).  Only rank 0 prints.  Returns the log of the series."""
    import time

    import numpy as np
    log = dict(epoch=[], elbo=[], recon_loss=[], epoch_time=[], steps=[])
    for epoch in range(nr_epochs):
        t0 = time.time()
        elbos, losses, sc = [], [], None
        for i, (lo, hi) in enumerate(train_batches):
            llo, lhi = shard_batch(lo, hi, ctx.world, ctx.rank, group_rows)
            if lhi <= llo:
                raise ValueError(f"batch [{lo}, {hi}) leaves rank {ctx.rank} of {ctx.world} without rows: use a batch size of at "
                                 f"least {ctx.world * group_rows} rows (also for the ragged last batch)")
            sc = local_step(llo, lhi, lo, hi, epoch, i)
            elbos.append(sc["elbo"]); losses.append(sc["recon_loss"])
            log["steps"].append(dict(epoch=epoch, rows=hi - lo, local_rows=lhi - llo, elbo=sc["elbo"],
                                     recon_loss=sc["recon_loss"], C_ma=sc["c_ma"], lagrange_mult=sc["lagrange"]))
        mse = float(np.sum(losses) / N_train)
        log["epoch"].append(epoch)
        log["elbo"].append(float(np.sum(elbos) if elbo_reduce == "sum" else np.mean(elbos)))
        log["recon_loss"].append(mse)
        log["epoch_time"].append(time.time() - t0)
        if ctx.rank == 0 and sc is not None:
            say(f"Epoch {epoch}: ELBO {elbo_reduce} {log['elbo'][-1]:.4f}  train MSE/px {mse:.6f}  "
                f"{log['epoch_time'][-1]:.2f}s  (C_ma {sc['c_ma']:.5f}, lagrange {sc['lagrange']:.4f})"
                + (f"  [{ctx.world} ranks]" if ctx.multi else ""))
        if on_epoch_end is not None:
            on_epoch_end(epoch, log)
    return log


class TorchDistComm:
    """Stand-in for engine.RcclComm over torch.distributed (backend nccl = RCCL on a GPU box): the exchange points of
    SpritesStepEngine.step when the library's own communicator cannot be created on every rank.  Same contract for
    `run(ops, stream)`: the collectives are enqueued on the engine's compute stream (torch orders its NCCL work after the
    current stream and makes the current stream wait for it)."""
    handle = None

    def __init__(self, group=None):
        import torch.distributed as dist
        self.group = group
        self.rank, self.world_size = dist.get_rank(group), dist.get_world_size(group)

    def _on(self, stream):
        import contextlib
        if stream is None:
            return contextlib.nullcontext()
        return torch.cuda.stream(stream if isinstance(stream, torch.cuda.Stream) else torch.cuda.ExternalStream(int(stream)))

    def run(self, ops, stream):
        from .engine import dist_exchange
        with self._on(stream):
            dist_exchange(ops, self.group)

    def all_reduce(self, tensor, stream):
        import torch.distributed as dist
        with self._on(stream):
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)

    def close(self):
        pass


def make_comm(ctx, device, timeout_s=180.0):
    """(communicator, reason): the library's RCCL communicator when every rank can create one, else a TorchDistComm and the
    reason of the fallback; (None, None) with a single rank and no forced distribution."""
    if not ctx.multi:
        return None, None
    comm, why = library_comm(True, ctx.local_rank, device, timeout_s=timeout_s)
    if comm is None:
        comm = TorchDistComm()
    return comm, why


def attach_library_comm(eng, ctx, timeout_s=180.0):
    """Gives the engine the library's RCCL communicator when every rank can create one (library_comm below); otherwise the engine
    keeps exchanging through torch.distributed between the phases.  Returns the reason of the fallback, or None."""
    if not ctx.multi:
        return None
    comm, why = library_comm(True, ctx.local_rank, eng.device, timeout_s=timeout_s)
    if comm is not None:
        eng.attach_comm(comm)
    return why


def library_comm(multi, local_rank, dev, timeout_s=180.0, make_id=None, make_comm=None):
    """The library's own RCCL communicator (engine.RcclComm), or (None, reason).  With several ranks the decision is COLLECTIVE
    and every collective of the decision is issued by the MAIN thread of every rank in the same order, whatever fails where:
      1. rank 0 creates the unique id (or None on failure) and broadcasts it -- all ranks take part, always;
      2. MIN over ranks of "I hold an id and my library loaded" -- a rank that cannot even load the library is seen here,
         before anybody enters ncclCommInitRank (which would otherwise wait for it);
      3. ONLY ncclCommInitRank (no torch.distributed call) runs in a helper thread with a time limit;
      4. MIN over ranks of "my communicator exists".
    A rank that fails or times out anywhere therefore never leaves the others in a collective it does not join
    (the id broadcast used to sit inside the helper thread, so a rank failing before it paired its all_reduce
    with the other ranks' broadcast).  `make_id` / `make_comm(rank, world, id)` are injectable for the CPU (gloo) test of the
    decision path; test hooks: SVGP_BENCH_FAIL_LIBCOMM=1 (every rank), SVGP_BENCH_FAIL_LIBCOMM_RANK=<r>[:id|:init] (one rank)."""
    if make_id is None or make_comm is None:
        from .engine import RcclComm
        make_id = make_id or RcclComm.unique_id
        make_comm = make_comm or RcclComm
    if not multi:
        try:
            return make_comm(0, 1, make_id()), None
        except Exception as e:
            return None, repr(e)
    import threading
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    hook_all = os.environ.get("SVGP_BENCH_FAIL_LIBCOMM") == "1"
    hook_rank, _, hook_where = os.environ.get("SVGP_BENCH_FAIL_LIBCOMM_RANK", "").partition(":")
    hook_me = hook_rank != "" and int(hook_rank) == rank
    on_gpu = torch.device(dev).type == "cuda"

    def vote(flag):
        ok = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        return float(ok.item()) >= 0.5

    # 1. the id: created by rank 0 on its main thread, broadcast by every rank's main thread
    why, box = None, [None]
    if rank == 0:
        try:
            if hook_all or (hook_me and hook_where in ("", "id")):
                raise RuntimeError("library communicator disabled by SVGP_BENCH_FAIL_LIBCOMM")
            box[0] = make_id()
        except Exception as e:
            why = repr(e)
    dist.broadcast_object_list(box, src=0)
    # 2. every rank holds an id and can reach its library
    mine = box[0] is not None
    if mine and rank != 0:
        try:
            if hook_all or (hook_me and hook_where in ("", "id")):
                raise RuntimeError("library communicator disabled by SVGP_BENCH_FAIL_LIBCOMM")
            make_id()                                  # loads the library and resolves RCCL on this rank (id discarded)
        except Exception as e:
            mine, why = False, repr(e)
    if not vote(mine):
        return None, why or "another rank has no unique id / library"
    # 3. ncclCommInitRank alone, time-limited
    res = {}

    def work():
        try:
            if on_gpu:
                torch.cuda.set_device(local_rank)      # the current device is per thread; ncclCommInitRank binds to it
            if hook_me and hook_where == "init":
                raise RuntimeError("library communicator disabled by SVGP_BENCH_FAIL_LIBCOMM_RANK")
            res["comm"] = make_comm(rank, world, box[0])
        except Exception as e:
            res["err"] = repr(e)

    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(timeout_s)
    why = "ncclCommInitRank timed out" if t.is_alive() else res.get("err")
    # 4. everybody has a communicator, or nobody uses one
    if not vote(res.get("comm") is not None and not t.is_alive()):
        # this rank may hold a communicator the vote has just discarded: release it; a helper thread still inside
        # ncclCommInitRank cannot be interrupted from here -- it is a daemon thread and is reported
        # (ncclCommDestroy can itself block while a peer is still inside ncclCommInitRank -- closed in a helper thread
        # with the same time limit, so the graceful fallback cannot turn into a hang)
        mine = res.get("comm")
        if mine is not None and hasattr(mine, "close"):
            cres = {}

            def close_it():
                try:
                    mine.close()
                except Exception as e:
                    cres["err"] = repr(e)

            ct = threading.Thread(target=close_it, daemon=True)
            ct.start()
            ct.join(min(timeout_s, 30.0))
            if ct.is_alive():
                why = f"{why or ''} (closing the discarded communicator timed out)".strip()
            elif "err" in cres:
                why = f"{why or ''} (closing the discarded communicator failed: {cres['err']})".strip()
        if t.is_alive():
            why = (why or "") + " [helper thread still inside ncclCommInitRank]"
        return None, why or "another rank has no communicator"
    return res["comm"], None
