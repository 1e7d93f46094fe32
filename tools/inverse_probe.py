"""Timing of the batched SPD inverse forms at the sizes the GP block uses (information; HIP events via torch):
fused Gauss-Jordan / potrf + potri (svgp_spd_inverse_batched) against potrf + potri, and torch.linalg.inv (rocSOLVER)."""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from svgp_vae_amd import _lib
lib = _lib.load_library()
DT = torch.float64
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2] * 1e3
out = {}
for m, batch in ((256, 17), (512, 16), (800, 65), (2048, 17)):
    g = torch.Generator(device="cuda").manual_seed(m)
    X = torch.randn(batch, m, m + 8, dtype=DT, device="cuda", generator=g)
    A = X @ X.transpose(1, 2) / m + 0.05 * torch.eye(m, dtype=DT, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ld = torch.zeros(batch, dtype=DT, device="cuda")
    w1 = torch.zeros(lib.svgp_spd_inverse_workspace_elems(m, batch), dtype=DT, device="cuda")
    w2 = torch.zeros(lib.svgp_potrf_workspace_elems(m, batch), dtype=DT, device="cuda")
    w3 = torch.zeros(lib.svgp_potri_workspace_elems(m, batch), dtype=DT, device="cuda")
    B = A.clone()
    def gj():
        B.copy_(A); _lib.call("svgp_spd_inverse_batched", m, batch, B.data_ptr(), ld.data_ptr(), w1.data_ptr(), s)
    def chol():
        B.copy_(A); _lib.call("svgp_potrf_batched", m, batch, B.data_ptr(), m, m * m, ld.data_ptr(), w2.data_ptr(), s)
    def cholinv():
        chol(); _lib.call("svgp_potri_batched", m, batch, B.data_ptr(), w2.data_ptr(), w3.data_ptr(), s)
    cp = timed(lambda: B.copy_(A))
    r = dict(copy_us=cp, spd_inverse_us=timed(gj) - cp, potrf_us=timed(chol) - cp, potrf_potri_us=timed(cholinv) - cp,
             torch_inv_us=timed(lambda: torch.linalg.inv(A)), torch_cholesky_us=timed(lambda: torch.linalg.cholesky(A)))
    r["inverse_TFLOPs_2m3"] = 2.0 * m ** 3 * batch / (r["spd_inverse_us"] * 1e-6) / 1e12
    out[f"m{m}_b{batch}"] = {k: round(v, 2) for k, v in r.items()}
print(json.dumps(out))
