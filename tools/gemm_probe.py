"""Development probe: svgp_dgemm_batched TFLOP/s at the large-m shapes, next to torch.bmm (rocBLAS) on the same inputs."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from svgp_vae_amd import _lib
DT = torch.float64
def run(ta, tb, M, N, K, batch, reps=5):
    A = torch.randn((batch, K, M) if ta else (batch, M, K), dtype=DT, device="cuda")
    B = torch.randn((batch, N, K) if tb else (batch, K, N), dtype=DT, device="cuda")
    Cm = torch.empty(batch, M, N, dtype=DT, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    f = lambda: _lib.call("svgp_dgemm_batched", ta, tb, M, N, K, 1.0, A.data_ptr(), A.shape[-1], A[0].numel(), B.data_ptr(),
                          B.shape[-1], B[0].numel(), 0.0, Cm.data_ptr(), N, M * N, batch, s)
    opA = A.transpose(1, 2) if ta else A
    opB = B.transpose(1, 2) if tb else B
    h = lambda: torch.bmm(opA, opB)
    out = []
    for fn in (f, h):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(2.0 * M * N * K * batch / (e0.elapsed_time(e1) / reps * 1e-3) / 1e12)
    err = float((Cm - h()).abs().max())
    print(f"ta={ta} tb={tb} M={M} N={N} K={K} batch={batch}: svgp {out[0]:.1f} TF  rocBLAS {out[1]:.1f} TF  maxerr {err:.2e}", flush=True)
for ta, tb in ((0, 0), (1, 0), (0, 1), (1, 1)):
    run(ta, tb, 800, 800, 800, 64)
run(0, 0, 256, 256, 256, 16); run(0, 0, 1024, 256, 256, 16); run(0, 0, 500, 800, 800, 64); run(1, 0, 800, 800, 500, 64)
run(0, 0, 2048, 2048, 2048, 16)
