"""Information-only timing of the moving-ball step (BASELINE configs[0] shape: 35 videos x 30 frames x 32x32, MLP 500):
HIP steps/s per --elbo choice (HIP events on the engine's stream, fresh device-synthesised batch per step as in the
reference) beside the CPU oracle (torch autograd, float64) on the host cores.

    python tools/ball_bench.py [--steps 200] [--cpu-steps 3] > profiles/rNN_ball.json"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--cpu-steps", type=int, default=3, help="0 skips the CPU oracle")
    ap.add_argument("--elbo", nargs="+", default=["VAE", "GPVAE_Pearce", "NP", "SVGPVAE_Hensman", "SVGPVAE_Titsias"])
    a = ap.parse_args()
    from svgp_vae_amd import BALL_experiment as BE, ball
    from oracle import ball_oracle as BO, pearce_vae_oracle as PO
    res = {}
    for elbo in a.elbo:
        # BASELINE configs[0] (`BALL_experiment.py --elbo VAE`, no --GP_joint: the length scale is the constant model_lt =
        # 0.001 of BALL_experiment.py:46-50); the GP variants run with the joint optimisation of the README command
        flags = ["--elbo", elbo, "--clip_qs", "--jitter", "1e-6"] + ([] if elbo == "VAE" else ["--GP_joint", "--ip_joint"])
        args = BE.build_parser().parse_args(flags)
        eng = BE.build_engine(args)
        src = ball.VideoBatchSource(tmax=30, px=32, py=32, lt=2, batch=35, seed=1, r=3)

        def one():
            v = src(); v.record_stream(eng.stream)
            eng.step(v, None, adam=True)
        for _ in range(10):
            one()
        eng.stream.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            one()
        eng.stream.synchronize(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # CPU oracle on the same shape
        DT = torch.float64
        g = torch.Generator().manual_seed(0)
        vid = PO.make_video_batch(tmax=30, px=32, py=32, lt=2.0, batch=35, r=3, generator=g, dtype=DT)
        p = {k: v.to(DT) for k, v in PO.init_mlp_params(32, 32, hidden=500, seed=0).items()}
        eps = torch.randn(35, 30, 2, dtype=DT, generator=g)
        t1 = time.perf_counter()
        for _ in range(a.cpu_steps):
            if elbo.startswith("SVGPVAE"):
                q = dict(p)
                for c in "xy":
                    q[f"ip_{c}"] = torch.linspace(1.0, 30.0, 15, dtype=DT); q[f"l_{c}"] = torch.tensor(2.0, dtype=DT)
                BO.loss_and_grads(q, vid, eps, beta=1.0, titsias="Titsias" in elbo, jitter=1e-6, clipping_qs=True)
            else:
                lt = 0.001 if elbo == "VAE" else 2.0
                q = dict(p); q["l_x"] = q["l_y"] = torch.tensor(lt, dtype=DT)
                ri = torch.stack([torch.randperm(30, generator=g) for _ in range(35)]) if elbo == "NP" else None
                BO.pearce_loss_and_grads(q, vid, eps, beta=1.0, type_elbo=elbo, lt=lt, ran_ind=ri, con_tf=15 if ri is not None else None)
        cpu = (time.perf_counter() - t1) / max(a.cpu_steps, 1) if a.cpu_steps else float("nan")
        res[elbo] = dict(hip_steps_per_s=a.steps / dt, hip_ms_per_step=1e3 * dt / a.steps, cpu_oracle_ms_per_step=1e3 * cpu,
                         cpu_threads=torch.get_num_threads(), elbo_after=eng.scalars()["elbo"])
    print(json.dumps(dict(workload="BASELINE configs[0] shape: batch 35, tmax 30, 32x32, MLP 500, m 15, float64; wall clock "
                                   "incl. the per-step device video synthesis", steps=a.steps, results=res)))


if __name__ == "__main__":
    main()
