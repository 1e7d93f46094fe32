"""Informational timing of the SPRITES step (BASELINE configs[3] shape): 500 frames per GPU (10 characters x 50),
L=64, L_action=8, L_character=16, m inducing points, jitter 0.01, cosine-normalised linear kernels, GECO, gradient
clip 1e6.  One GPU: `python tools/sprites_bench.py [m]`.  N GPUs (weak scaling, 500 frames per GPU, whole character
groups per rank, three in-library RCCL all-reduces per step): `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/sprites_bench.py [m]`.  `--force-comm` runs the
data-parallel schedule with a 1-rank communicator."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from svgp_vae_amd import sprites as S
from svgp_vae_amd.engine import RcclComm
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
m = int(argv[0]) if argv else 800
rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
torch.cuda.set_device(local)
comm = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    comm = RcclComm.from_process_group()
elif "--force-comm" in sys.argv:
    comm = RcclComm(0, 1, RcclComm.unique_id())
b, frames, L, La, Lc, n_act = 500, 50, 64, 8, 16, 72
rs = np.random.RandomState(0)
svgp = S.spritesSVGP(False, False, rs.normal(0, 1.5, (m, La + Lc)), 'main', 0.01, 50000, La, rs.normal(0, 1.5, (n_act, La)),
                     Lc, L, K_obj_normalize=True)
eng = S.SpritesStepEngine(S.spritesVAE(L), S.sprites_representation_network(Lc), svgp, b_max=b, seg_len=frames, geco=True,
                          kappa_squared=0.0075, clip_grad=1e6, device=f"cuda:{local}", rank=rank, world_size=world, comm=comm)
dev = eng.dev
g = torch.Generator().manual_seed(rank)
img = torch.rand(b, 64, 64, 3, dtype=torch.float64, generator=g).to(dev)
ids = torch.tensor(np.random.RandomState(rank).randint(0, n_act, b), dtype=torch.float64, device=dev)
for _ in range(3):
    eng.step(img, ids, None, adam=True)
eng.stream.synchronize(); torch.cuda.synchronize()
if world > 1:
    dist.barrier()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    eng.step(img, ids, None, adam=True)
eng.stream.synchronize(); torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt = (time.perf_counter() - t0) / n
if world > 1:
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
sc = eng.scalars()
if rank == 0:
    import ctypes
    ctypes.CDLL(None).fflush(None)
    print(json.dumps({"workload": f"SPRITES step, {b} frames per GPU, L={L} m={m} float64", "n_gpus": world,
                      "ms_per_step": dt * 1e3, "steps_per_s": 1 / dt, "frames_per_s": world * b / dt,
                      "exchange": None if comm is None else "in-library RCCL all-reduce x3 (statA, statB, grad+sums)",
                      "elbo": sc["elbo"], "recon_loss": sc["recon_loss"], "finite": bool(np.isfinite(sc["elbo"]))}))
if world > 1:
    dist.destroy_process_group()
