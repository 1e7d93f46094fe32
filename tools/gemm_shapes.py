"""Tuning aid: svgp_dgemm_batched on a list of shapes given as ta,tb,M,N,K,batch (sustained rate over 20 launches)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from svgp_vae_amd import _lib
DT = torch.float64
for spec in sys.argv[1:]:
    ta, tb, M, N, K, batch = (int(v) for v in spec.split(","))
    A = torch.randn((batch, K, M) if ta else (batch, M, K), dtype=DT, device="cuda")
    B = torch.randn((batch, N, K) if tb else (batch, K, N), dtype=DT, device="cuda")
    Cm = torch.empty(batch, M, N, dtype=DT, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: _lib.call("svgp_dgemm_batched", ta, tb, M, N, K, 1.0, A.data_ptr(), A.shape[-1], A[0].numel(), B.data_ptr(),
                          B.shape[-1], B[0].numel(), 0.0, Cm.data_ptr(), N, M * N, batch, st)
    best = 1e9
    for _ in range(3):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e-3)
    print(f"ta={ta} tb={tb} {M:5d} {N:5d} {K:5d} x {batch}: {best*1e6:8.1f} us  {2.0*M*N*K*batch/best/1e12:5.1f} TF", flush=True)
