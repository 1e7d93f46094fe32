"""bench.py's collective communicator bootstrap (`library_comm`) with real processes (gloo, world size 2 and 3) and
injected id / communicator factories: whatever fails on whichever rank — the id on rank 0, the library on ONE other rank,
ncclCommInitRank raising on one rank, ncclCommInitRank hanging on one rank — every rank returns, every rank takes the
same decision, and the process group is still usable afterwards (the id broadcast used to run in the helper
thread, so a rank that failed before it paired an all_reduce with the other ranks' broadcast)."""
import os
import socket
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, scenario, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bad = world - 1                                  # the misbehaving rank (never rank 0 except for "id0")

        def make_id():
            if scenario == "id0" and rank == 0:
                raise RuntimeError("no id")
            if scenario == "lib1" and rank == bad:
                raise OSError("library not loadable on this rank")
            return b"x" * 128

        def make_comm(r, w, uid):
            assert uid == b"x" * 128 and (r, w) == (rank, world)
            if scenario == "init_raises" and rank == bad:
                raise RuntimeError("ncclCommInitRank failed")
            if scenario == "init_hangs" and rank == bad:
                time.sleep(3600)
            return ("comm", r)

        t0 = time.time()
        comm, why = bench.library_comm(True, 0, "cpu", timeout_s=3.0, make_id=make_id, make_comm=make_comm)
        dt = time.time() - t0
        # the group must still be in step: one more collective with a known answer
        x = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(x)
        ret.put((rank, comm, why, float(x.item()), dt))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("scenario", ["ok", "id0", "lib1", "init_raises", "init_hangs"])
def test_library_comm_decision_is_collective(scenario, world):
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scenario, ret), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(ret.get() for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[0] for g in got] == list(range(world))
    for rank, comm, why, total, dt in got:
        assert total == world * (world + 1) / 2          # collectives still paired after the decision
        assert dt < 30
        if scenario == "ok":
            assert comm == ("comm", rank) and why is None
        else:
            assert comm is None and why                  # every rank falls back, with a reason
    if scenario == "lib1":
        assert "library not loadable" in got[world - 1][2]
    if scenario == "init_hangs":
        assert "timed out" in got[world - 1][2]


def test_env_hook_fails_one_rank_only(monkeypatch):
    """SVGP_BENCH_FAIL_LIBCOMM_RANK=<r>:init — the hook the GPU-side bench test uses — goes through the same votes."""
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    port = _free_port()
    monkeypatch.setenv("SVGP_BENCH_FAIL_LIBCOMM_RANK", "1:init")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, "ok", ret), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(ret.get() for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(g[1] is None and g[2] for g in got) and "SVGP_BENCH_FAIL_LIBCOMM_RANK" in got[1][2]
