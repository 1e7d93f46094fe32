"""One batched float64 GEMM shape, a few launches, for rocprofv3 --pmc / --kernel-trace runs: gemm_one.py M N K batch [ta tb]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from svgp_vae_amd import _lib
M, N, K, batch = (int(a) for a in sys.argv[1:5])
ta, tb = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (0, 0)
DT = torch.float64
A = torch.randn((batch, K, M) if ta else (batch, M, K), dtype=DT, device="cuda")
B = torch.randn((batch, N, K) if tb else (batch, K, N), dtype=DT, device="cuda")
Cm = torch.empty(batch, M, N, dtype=DT, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(6):
    _lib.call("svgp_dgemm_batched", ta, tb, M, N, K, 1.0, A.data_ptr(), A.shape[-1], A[0].numel(), B.data_ptr(), B.shape[-1],
              B[0].numel(), 0.0, Cm.data_ptr(), N, M * N, batch, s)
torch.cuda.synchronize()
