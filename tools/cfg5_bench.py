"""Config 5: synthetic N-row / m = 2048 / L = 16 float32 statistics stress on ONE GPU's shard
(131 072 rows = 2^20 / 8).  Times the K_nm build (HBM-write-bound) and the S_l / v_l pass (fp32 MFMA-bound)
with HIP events, checks a random-probe identity in float64, prints one JSON object.
Under torchrun (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
tools/cfg5_bench.py`) every rank holds its own 131 072-row shard and the statistics pass ends with the in-library RCCL
all-reduce of S (L, m, m) and v; rank 0 prints, `stats.ms` then includes the all-reduce."""
import argparse, json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from svgp_vae_amd import stream_stats as SS

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=131072)
ap.add_argument("--m", type=int, default=2048)
ap.add_argument("--L", type=int, default=16)
ap.add_argument("--M", type=int, default=8)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--kind", choices=["periodic", "se"], default="periodic")
args = ap.parse_args()
rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
torch.cuda.set_device(local)
comm = None
if world > 1:
    import torch.distributed as dist
    from svgp_vae_amd.engine import RcclComm
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    comm = RcclComm.from_process_group()
dev = torch.device(f"cuda:{local}")
g = torch.Generator(device="cpu").manual_seed(rank)
n, m, L, M = args.n, args.m, args.L, args.M
if args.kind == "periodic":
    n_obj = 400
    kd = SS.kernel_desc(SS.PERIODIC_LINEAR, 2, M, n_table=n_obj, params=(1.0, 1.0))
    tab = (torch.randn(n_obj, M, generator=g) * 1.5).to(dev)
    x = torch.cat([torch.randint(0, n_obj, (n, 1), generator=g).float(), torch.rand(n, 1, generator=g) * 6.2832,
                   torch.randn(n, M, generator=g)], 1).to(dev).contiguous()
    z = torch.cat([torch.zeros(m, 1), torch.rand(m, 1, generator=g) * 6.2832, torch.randn(m, M, generator=g) * 1.5], 1).to(dev).contiguous()
    D = 2 + M
else:
    kd = SS.kernel_desc(SS.SE_SE, 8, 16, n_table=72, params=(5.0, 1.0, 7.0, 1.0))
    tab = (torch.randn(72, 8, generator=g) * 1.5).to(dev)
    x = torch.cat([torch.randint(0, 72, (n, 1), generator=g).float(), torch.randn(n, 16, generator=g) * 1.5], 1).to(dev).contiguous()
    z = (torch.randn(m, 24, generator=g) * 1.5).to(dev).contiguous()
    D = 24
means = torch.randn(n, L, generator=g).to(dev)
vars_ = (torch.rand(n, L, generator=g) * 9.999 + 1e-3).to(dev)

fr = SS.features(kd, x, inducing=False, table=tab)
fi = SS.features(kd, z, inducing=True)
K = torch.empty((n, m), dtype=torch.float32, device=dev)
ws = SS.stats_workspace(n, m, L, dev)
S = torch.empty((L, m, m), dtype=torch.float32, device=dev)
v = torch.empty((L, m), dtype=torch.float32, device=dev)

def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2] * 1e-3

t_feat = timed(lambda: SS.features(kd, x, inducing=False, table=tab), args.reps)
t_knm = timed(lambda: SS.knm(kd, fr, n, fi, m, out=K), args.reps)
t_stats = timed(lambda: SS.stats(K, means, vars_, ws=ws, S=S, v=v, comm=comm), args.reps)
if world > 1:
    # the probe below is a single-shard identity: recompute this rank's statistics without the exchange
    SS.stats(K, means, vars_, ws=ws, S=S, v=v)
    torch.cuda.synchronize()

# probe: for a random w, S_l w == K^T (p_l * (K w)) in float64 (GEMV chain on the GPU, not the MFMA kernel)
w = torch.randn(m, generator=g).to(dev).double()
Kd_w = (K.double() @ w)
p = torch.where(vars_ == 0, torch.zeros_like(vars_), 1.0 / vars_).double()
err = 0.0
for l in range(L):
    ref = K.double().t() @ (p[:, l] * Kd_w) if n * m <= 2 ** 28 else K.t().double() @ (p[:, l] * Kd_w)
    got = S[l].double() @ w
    err = max(err, float((got - ref).abs().max() / ref.abs().max()))
vref = K.double().t() @ (p * means.double())
verr = float((v.double().t() - vref).abs().max() / vref.abs().max())
sym = float((S - S.transpose(1, 2)).abs().max())

alg_flops = float(L) * n * m * m          # symmetric count (L N m^2)
knm_bytes = 4.0 * (n * m + n * D + m * D)
if world > 1:
    tt = torch.tensor([t_feat, t_knm, t_stats], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t_feat, t_knm, t_stats = (float(x) for x in tt)
if rank == 0:
  import ctypes
  ctypes.CDLL(None).fflush(None)
  print(json.dumps({
    "config": f"cfg5: {world} shard(s) of n={n} m={m} L={L} kind={args.kind} D={D} float32",
    "n_gpus": world, "whole_job_TFLOPs": world * float(L) * n * m * m / t_stats / 1e12,
    "features_ms": t_feat * 1e3,
    "knm_build": {"ms": t_knm * 1e3, "algorithmic_bytes": knm_bytes, "GBps": knm_bytes / t_knm / 1e9,
                  "frac_of_8TBps": knm_bytes / t_knm / 8e12},
    "stats": {"ms": t_stats * 1e3, "algorithmic_flops": alg_flops, "TFLOPs": alg_flops / t_stats / 1e12,
              "frac_of_157TF": alg_flops / t_stats / 157.3e12, "executed_over_algorithmic": 2.0 * (m // 256 * (m // 256 + 1) / 2) / (m / 256) ** 2 if m % 256 == 0 else None},
    "probe_rel_err_S": err, "probe_rel_err_v": verr, "max_asym": sym}))
if world > 1:
    dist.destroy_process_group()
