import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from svgp_vae_amd import _lib
lib = _lib.load_library()
DT = torch.float64
for m, batch in ((128, 2), (128, 1), (192, 1)):
    g = torch.Generator(device="cuda").manual_seed(m)
    X = torch.randn(batch, m, m + 8, dtype=DT, device="cuda", generator=g)
    A = X @ X.transpose(1, 2) / m + 0.05 * torch.eye(m, dtype=DT, device="cuda")
    Lf = A.clone()
    ld = torch.zeros(batch, dtype=DT, device="cuda")
    work = torch.zeros(lib.svgp_potrf_workspace_elems(m, batch), dtype=DT, device="cuda")
    _lib.call("svgp_potrf_batched", m, batch, Lf.data_ptr(), m, m * m, ld.data_ptr(), work.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = torch.linalg.cholesky(A)
    bad = torch.isnan(Lf).nonzero()
    print(m, batch, "nan count", bad.shape[0], "first", bad[:5].tolist(), "err", float((torch.nan_to_num(Lf) - want).abs().max()), "ld", ld.tolist())
    nb = (m + 63) // 64
    Linv = work[:batch * nb * 4096].view(batch, nb, 64, 64)
    print("  Linv nan", int(torch.isnan(Linv).sum()))
