"""Sustained rate of svgp_dgemm_batched next to torch.bmm (rocBLAS) on the shapes of configs 3 / SPRITES m = 800 / config 5:
30 back-to-back launches each (5-launch samples read 10 % lower: clocks).  usage: gemm_sweep.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from svgp_vae_amd import _lib
SHAPES = [(1, 0, 256, 256, 1024, 16), (0, 1, 1024, 256, 256, 16), (0, 0, 256, 256, 256, 16), (0, 1, 256, 256, 256, 16),
          (0, 0, 800, 800, 800, 64), (1, 0, 800, 800, 800, 64), (0, 1, 800, 800, 800, 64), (1, 1, 800, 800, 800, 64),
          (1, 0, 800, 800, 500, 64), (0, 1, 500, 800, 800, 64), (0, 0, 500, 800, 800, 64),
          (0, 0, 2048, 2048, 2048, 16), (0, 1, 2048, 2048, 2048, 16), (1, 0, 2048, 2048, 2048, 16),
          (0, 0, 72, 72, 72, 64), (1, 0, 72, 72, 512, 64), (0, 1, 512, 72, 72, 64)]
DT = torch.float64
REPS = 30
for ta, tb, M, N, K, batch in SHAPES:
    A = torch.randn((batch, K, M) if ta else (batch, M, K), dtype=DT, device="cuda")
    B = torch.randn((batch, N, K) if tb else (batch, K, N), dtype=DT, device="cuda")
    Cm = torch.empty(batch, M, N, dtype=DT, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: _lib.call("svgp_dgemm_batched", ta, tb, M, N, K, 1.0, A.data_ptr(), A.shape[-1], A[0].numel(), B.data_ptr(),
                          B.shape[-1], B[0].numel(), 0.0, Cm.data_ptr(), N, M * N, batch, st)
    opA = A.transpose(1, 2) if ta else A
    opB = B.transpose(1, 2) if tb else B
    h = lambda: torch.bmm(opA, opB)
    us = []
    for fn in (f, h, f, h):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS): fn()
        e1.record(); torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3 / REPS)
    u0, u1 = min(us[0], us[2]), min(us[1], us[3])
    tf = lambda u: 2.0 * M * N * K * batch / u * 1e-6
    print(f"ta={ta} tb={tb} {M:5d} {N:5d} {K:5d} x{batch:3d}: svgp {u0:8.1f} us {tf(u0):5.1f} TF   rocBLAS {u1:8.1f} us {tf(u1):5.1f} TF", flush=True)
