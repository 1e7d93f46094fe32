"""One batched SPD inverse shape for rocprofv3 runs: inverse_one.py m batch [reps]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from svgp_vae_amd import _lib
lib = _lib.load_library()
m, batch = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
DT = torch.float64
g = torch.Generator(device="cuda").manual_seed(m)
X = torch.randn(batch, m, m + 8, dtype=DT, device="cuda", generator=g)
A = X @ X.transpose(1, 2) / m + 0.05 * torch.eye(m, dtype=DT, device="cuda")
s = torch.cuda.current_stream().cuda_stream
ld = torch.zeros(batch, dtype=DT, device="cuda")
w = torch.zeros(lib.svgp_spd_inverse_workspace_elems(m, batch), dtype=DT, device="cuda")
B = A.clone()
for _ in range(reps):
    B.copy_(A)
    _lib.call("svgp_spd_inverse_batched", m, batch, B.data_ptr(), ld.data_ptr(), w.data_ptr(), s)
torch.cuda.synchronize()
print("residual", float((B @ A - torch.eye(m, dtype=DT, device="cuda")).abs().max()))
