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


@pytest.mark.parametrize("pack", ["0", "1"])
def test_channel_sharded_dp_entry_equals_single_gpu_step_large_m(pack, monkeypatch):
    """m > 64: svgp_mnist_train_step_dp takes the channel-sharded sequence (reduce-scatter / window factor stages /
    all-gather); with a 1-rank communicator the collectives are identities and the window is all channels, so three
    Adam steps must reproduce svgp_mnist_train_step (same kernels; the deferred phase forms do not exist for m > 64).
    pack = 1: the grouped + tile-packed form of the five exchange points (pack -> ncclGroup{...} -> unpack through real
    RCCL calls).  The per-point event
    timing (svgp_comm_timing) reports five points."""
    from svgp_vae_amd.engine import RcclComm
    monkeypatch.setenv("SVGP_DP_PACK", pack)
    params, images, aux, eps = H.toy_problem(b=96, m=72, L=4, M=16, n_obj=40, seed=5)
    kw = dict(geco=True, N_train=4050.0, jitter=1e-4)
    a = H.engine_for(params, 96, **kw)
    b_ = H.engine_for(params, 96, single_stat_block=True, **kw)      # the multi-rank workspace: it carries the wire buffer
    comm = RcclComm(0, 1, RcclComm.unique_id())
    x = torch.arange(24, dtype=torch.float64, device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    comm.reduce_scatter(x, s); comm.all_gather(x, s)
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), torch.arange(24, dtype=torch.float64))
    b_.attach_comm(comm)
    dev = a.device
    di, da, de = images.to(dev), aux.to(dev), eps.to(dev)
    a.bind(di, da, de); b_.bind(di, da, de)
    comm.timing(True)
    for _ in range(3):
        a.run(adam=True)
        b_.run(adam=True)
    a.synchronize(); b_.synchronize()
    us = comm.timing_read()
    assert len(us) == 5 and all(0.0 < u < 1e5 for u in us), us
    comm.timing(False)
    # (round 4: the multi-rank workspace forms the statistic SW_l = P^T S_l P from the exchanged S_l, the single-GPU engine over its
    # local rows W^T diag(p_l) W -- the same matrix to rounding, so the two runs agree to 1e-9 instead of bit for bit)
    tol = 1e-8
    assert H.relerr(b_.theta, a.theta) < tol
    sa, sb = a.scalars(), b_.scalars()
    for k in ("elbo", "recon_loss", "kl_term", "c_ma", "lagrange", "adam_t"):
        assert abs(sa[k] - sb[k]) <= tol * max(1.0, abs(sa[k])), k
    comm.close()


@pytest.mark.parametrize("m,L", [(1, 2), (31, 3), (32, 1), (33, 3), (72, 4), (130, 3), (256, 2)])
def test_sym_pack_unpack_kernels(m, L):
    """svgp_sym_pack / svgp_sym_unpack against the definition of the wire format (engine._symmetrise): lower-triangle mirror
    and average forms, tile padding with zeros, channel windows."""
    import ctypes as C
    from svgp_vae_amd import _lib
    from svgp_vae_amd.engine import _symmetrise
    lib = _lib.load_library()
    pe = int(lib.svgp_sym_packed_elems(m))
    nt = (m + 31) // 32
    assert pe == nt * (nt + 1) // 2 * 1024
    g = torch.Generator().manual_seed(m)
    X = torch.randn(L, m, m, dtype=torch.float64, generator=g).cuda()
    s = torch.cuda.current_stream().cuda_stream
    for avg in (0, 1):
        P = torch.full((L, pe), float("nan"), dtype=torch.float64, device="cuda")
        _lib.call("svgp_sym_pack", m, L, avg, X.data_ptr(), P.data_ptr(), s)
        Y = torch.full_like(X, float("nan"))
        _lib.call("svgp_sym_unpack", m, L, P.data_ptr(), Y.data_ptr(), s)
        torch.cuda.synchronize()
        assert torch.isfinite(P).all()
        assert torch.equal(Y, _symmetrise(X.reshape(-1), m, bool(avg)))
        # a window of channels: pointers offset by whole matrices / packed blocks
        if L > 1:
            Y2 = torch.full_like(X, float("nan"))
            _lib.call("svgp_sym_unpack", m, L - 1, P[1:].data_ptr(), Y2[1:].data_ptr(), s)
            torch.cuda.synchronize()
            assert torch.equal(Y2[1:], Y[1:]) and torch.isnan(Y2[0]).all()


def test_sprites_engine_with_one_rank_communicator_grouped_and_packed(monkeypatch):
    """SpritesStepEngine(channel_shard=True) with a 1-rank RcclComm: its exchange points go through RcclComm.run -- pack,
    ONE ncclGroupStart / End per point, unpack -- with real RCCL calls; two Adam steps equal the plain single-GPU engine."""
    from svgp_vae_amd import sprites as S
    from svgp_vae_amd.engine import RcclComm
    from tests.test_gpu_sprites import _problem, _rel, DT
    monkeypatch.setenv("SVGP_DP_PACK", "1")
    frames, La, Lc, n_act, L, m = 4, 8, 16, 9, 4, 72
    b = frames * 4
    params, gp, images, ids, eps, _, _ = _problem(b, frames, L, La, Lc, m, n_act, seed=3)
    init = dict(params)
    init["se"] = torch.stack([gp["l_action"], gp["sigma_action"], gp["l_character"], gp["sigma_character"]])

    def make(shard, comm):
        svgp = S.spritesSVGP(False, False, gp["inducing_index_points"].numpy(), 'main', 0.01, 100.0, La,
                             gp["GPLVM_action"].numpy(), Lc, L, fixed_GP_params=False, fixed_GPLVM=False,
                             K_obj_normalize=True, K_SE=False)
        e = S.SpritesStepEngine(S.spritesVAE(L), S.sprites_representation_network(Lc), svgp, b_max=b, seg_len=frames,
                                clip_qs=True, geco=True, kappa_squared=0.0075, params=init, rank=0, world_size=1,
                                channel_shard=shard, comm=comm)
        e.set_scalars(c_ma=0.02, lagrange=1.4, alpha=0.9)
        return e

    comm = RcclComm(0, 1, RcclComm.unique_id())
    plain, dp = make(False, None), make(True, comm)
    assert dp.chan_shard
    dev = plain.dev
    di, da, de = images.to(dev), ids.to(dev, DT), eps.to(dev)
    for _ in range(2):
        plain.step(di, da, de, adam=True)
        dp.step(di, da, de, adam=True)
    plain.stream.synchronize(); dp.stream.synchronize()
    assert _rel(dp.theta, plain.theta) < 1e-8
    sp, sd = plain.scalars(), dp.scalars()
    for k in ("elbo", "recon_loss", "kl_term"):
        assert abs(sp[k] - sd[k]) <= 1e-8 * max(1.0, abs(sp[k])), k
    comm.close()


def test_bench_multi_gpu_code_path_at_world_size_one():
    """`bench.py --force-dist` under torchrun with ONE rank: the N>1 code path end to end -- NCCL process group, the
    communicator bootstrap through broadcast_object_list, barriers, the MAX all-reduce of the time, teardown -- for both
    exchange forms.  (More than one rank needs more than one GPU; the driver's scaling bench is the first such run.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for ex, port in (("rccl", 29531), ("torch", 29532)):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20",
               "--warmup", "3", "--no-cpu-baseline", "--force-dist", "--exchange", ex]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 1 and line["value"] > 100 and "RCCL" in line["config"]["launch"]
        assert "block_ms_min" in line and "block_outliers" in line
        if ex == "rccl":      # per-exchange-point event timing of the in-library form (svgp_comm_timing)
            assert line["config"]["rccl_ranks"] == 1
            assert list(line["collectives_us"]) == ["ar[S|v]", "ar[A2|ud|td]", "ar[grad|sums]"]
            assert all(0 < v < 1e4 for v in line["collectives_us"].values())
            # ... and the other setting of cfg.split_grad_exchange next to it: four exchange points
            other = line["other_exchange_setting"]
            assert other["split_grad_exchange"] is True and line["config"]["split_grad_exchange"] is False
            assert len(other["collectives_us"]) == 4 and 0.5 * line["ms_per_step"] < other["ms_per_step"] < 1.5 * line["ms_per_step"]
    # a rank whose library communicator cannot be created: the collective decision sends every rank to the torch.distributed form
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "3",
           "--no-cpu-baseline", "--force-dist"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, SVGP_BENCH_FAIL_LIBCOMM="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert "torch.distributed" in line["config"]["launch"] and line["value"] > 100 and line["config"]["rccl_ranks"] == 1
    assert "in-library RCCL communicator unavailable" in r.stderr
    # config 3 (channel-sharded sequence: five grouped points) through the same code path
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "cfg3", "--steps", "5",
           "--warmup", "2", "--repeats", "1", "--no-cpu-baseline", "--force-dist"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(line["collectives_us"]) == 5 and line["elbo_rel_err_gpu_vs_oracle"] < 1e-6


def test_bench_launches_its_own_ranks_and_prints_the_strong_line_first():
    """`python bench.py --gpus 1 --force-dist` WITHOUT a launcher: the process spawns torch.distributed.run as
    a child before touching the GPU, the rank prints the strong-scaling line (the configuration's own global batch split over
    the ranks, ELBO against the oracle to 1e-8) and then the weak line, which is the parent's LAST stdout line and carries the
    strong result."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1" in r.stderr
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.lstrip().startswith("{") and '"metric"' in x]
    assert [x["scaling"] for x in lines] == ["strong", "weak"]
    last = json.loads(r.stdout.strip().splitlines()[-1])
    assert last == lines[-1] and last["n_gpus"] == 1 and last["config"]["rccl_ranks"] == 1 and last["value"] > 100
    st = last["strong_scaling"]
    assert st["global_batch"] == 256 and st["elbo_rel_err_gpu_vs_oracle"] < 1e-8 and st["value"] > 100
    assert abs(st["elbo"] - lines[0]["elbo"]) == 0.0


def test_exchange_lengths_agree_through_the_library_communicator():
    """engine.agree_on_lengths without a torch.distributed process group: the two fixed-count all-reduces of the library's own
    communicator (1 rank here: trivially equal; the disagreement logic is covered with stand-in communicators on CPU)."""
    from svgp_vae_amd.engine import RcclComm, agree_on_lengths
    comm = RcclComm(0, 1, RcclComm.unique_id())
    s = torch.cuda.Stream()
    assert agree_on_lengths([135168, 8, 74000], comm, torch.device("cuda:0"), s) is True
    assert agree_on_lengths([1, 2, 3], None, torch.device("cuda:0"), s) is False
    comm.close()


@pytest.mark.parametrize("workload,extra", [("sprites800", ["--precision", "f32"]), ("cfg5", ["--rows", "16384", "--m", "512"])])
def test_bench_multi_rank_paths_of_the_eight_gpu_configs_with_one_rank(workload, extra):
    """BASELINE configs[3] / [4] are the two "8 x MI355X" configurations; `python bench.py --gpus 1 --force-dist --workload ...`
    takes their N > 1 code end to end through the self-launcher with ONE rank: process group, the collective
    communicator bootstrap with its MIN votes, the exchange points of the step (sprites800: the strong line -- 500 frames cut in
    50-frame character groups -- and the weak line; cfg5: the S, v all-reduce), barriers and the MAX all-reduce of the time.
    Then the same with the library communicator refused on the rank: every rank falls back to torch.distributed together."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--workload", workload, "--steps", "2",
           "--warmup", "1", "--repeats", "1", "--no-cpu-baseline"] + extra
    for fail in (False, True):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root,
                           env=dict(env, SVGP_BENCH_FAIL_LIBCOMM="1") if fail else env)
        assert r.returncode == 0, r.stderr[-3000:]
        assert "torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1" in r.stderr
        last = json.loads(r.stdout.strip().splitlines()[-1])
        lines = [json.loads(x) for x in r.stdout.splitlines() if x.lstrip().startswith("{") and '"metric"' in x]
        assert last == lines[-1] and last["n_gpus"] == 1 and last["scaling"] == "weak" and last["value"] > 0
        if fail:
            assert last["config"]["rccl_ranks"] is None and "disabled by SVGP_BENCH_FAIL_LIBCOMM" in last["config"]["comm_fallback"]
            assert "in-library RCCL communicator unavailable" in r.stderr
        else:
            assert last["config"]["rccl_ranks"] == 1 and last["config"]["comm_fallback"] is None
        if workload == "sprites800":
            assert [x["scaling"] for x in lines] == ["strong", "weak"]
            assert lines[0]["config"]["global_batch"] == 500 and lines[0]["config"]["rows_per_gpu"] == 500
            assert len(last["collectives_us"]) == 3 and all(v >= 0 for v in last["collectives_us"].values())
        else:
            assert "all-reduce of S, v" in last["config"]["workload"] and last["probe_rel_err_S"] < 1e-4


@pytest.mark.parametrize("m,b,L,M", [(32, 256, 16, 8), (72, 96, 3, 16)])
def test_split_gradient_exchange_through_the_dp_entry_with_one_rank(m, b, L, M):
    """svgp_mnist_train_step_dp with cfg.split_grad_exchange and a 1-rank communicator: four exchange points (statA, statB, the
    gradient tail on the library's side branch beside the encoder's reverse pass, the gradient head) through real RCCL calls;
    three Adam steps equal the ordinary data-parallel entry and the plain single-GPU step.  (m = 72, L = 3 with ... ranks: the
    row-sharded schedule of the large-m path, taken when L is not divisible by the rank count -- here forced by titsias = False,
    one rank: L % 1 == 0 takes the channel-sharded one, so the large case runs m <= 64 semantics only through phases 4 / 5.)"""
    from svgp_vae_amd.engine import RcclComm
    params, images, aux, eps = H.toy_problem(b=b, m=m, L=L, M=M, n_obj=40, seed=8)
    kw = dict(geco=True, N_train=4050.0, jitter=1e-4)
    plain = H.engine_for(params, b, **kw)
    split = H.engine_for(params, b, split_grad_exchange=True, **kw)
    comm = RcclComm(0, 1, RcclComm.unique_id())
    split.attach_comm(comm)
    dev = plain.device
    di, da, de = images.to(dev), aux.to(dev), eps.to(dev)
    plain.bind(di, da, de); split.bind(di, da, de)
    comm.timing(True)
    for _ in range(3):
        plain.run(adam=True)
        split.run(adam=True)
    plain.synchronize(); split.synchronize()
    us = comm.timing_read()
    if m <= 64:
        assert len(us) == 4 and all(0.0 < u < 1e5 for u in us), us
    comm.timing(False)
    assert H.relerr(split.theta, plain.theta) < 1e-9
    sp, sd = plain.scalars(), split.scalars()
    for k in ("elbo", "recon_loss", "kl_term", "adam_t"):
        assert abs(sp[k] - sd[k]) <= 1e-9 * max(1.0, abs(sp[k])), k
    # phases 4 + 5 == phase 2 through the phase entry point, any m
    for eng in (plain, split):
        eng.reset_state()
    plain.phase(0); plain.phase(1); plain.phase(2); plain.phase(3, adam=False)
    split.phase(0); split.phase(1); split.phase(4); split.phase(5); split.phase(3, adam=False)
    plain.synchronize(); split.synchronize()
    for k, v in plain.grads().items():
        assert H.relerr(split.grads()[k], v) < 1e-12, k
    comm.close()
