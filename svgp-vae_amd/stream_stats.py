"""Full-data GP statistics in float32, streamed over N on the GPU (include/svgpvae_hip.h `svgp_stream_*`).

Host side of the conditional-generation statistics pass: the K_nm build and the per-channel products of
`precompute_GP_params_SVGPVAE` (SVGPVAE_model.py:989-1023) at N-sized inputs (SURVEY 8d config 5 /
8f rank 1).  All tensors are float32 CUDA tensors, row-major; every kernel is in libsvgpvae_hip.so and
runs on torch's current stream.  There is no CPU path.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import StreamKdesc, call

PERIODIC_LINEAR, LINEAR_LINEAR, SE_SE = 0, 1, 2


def kernel_desc(kind, d1, d2, *, normalize=False, n_table=0, params=()):
    """kind PERIODIC_LINEAR: d1 = 2, d2 = M, params = (l_GP, amplitude)       [mnistSVGP :416-417]
       kind LINEAR_LINEAR : d1 = L_action, d2 = L_character                  [spritesSVGP :547-548]
       kind SE_SE         : params = (l1, sigma1, l2, sigma2)                 [spritesSVGP :530-544]"""
    kd = StreamKdesc(kind=kind, d1=d1, d2=d2, normalize=int(normalize), n_table=n_table)
    for i, v in enumerate(params):
        kd.p[i] = float(v)
    return kd


def _check(t, name):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise _lib.SvgpError(f"{name} must be a contiguous float32 CUDA tensor")


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def features(kd, x, *, inducing, table=None):
    """Feature rows (n, d1+d2) followed by the per-row scalar (n), as one flat float32 tensor."""
    _lib.load_library()
    _check(x, "x")
    if table is not None:
        _check(table, "table")
    n = x.shape[0]
    out = torch.empty(_lib.load_library().svgp_stream_feature_elems(C.byref(kd), n), dtype=torch.float32, device=x.device)
    call("svgp_stream_features_f32", C.byref(kd), n, x.data_ptr(), x.shape[1], int(inducing),
         table.data_ptr() if table is not None else None, out.data_ptr(), _stream(x))
    return out


def knm(kd, feat_rows, n, feat_inducing, m, out=None):
    """K_nm (n, m) float32, materialised (HBM-write-bound)."""
    _check(feat_rows, "feat_rows"); _check(feat_inducing, "feat_inducing")
    if out is None:
        out = torch.empty((n, m), dtype=torch.float32, device=feat_rows.device)
    _check(out, "out")
    assert out.shape == (n, m)
    call("svgp_stream_knm_f32", C.byref(kd), n, m, feat_rows.data_ptr(), feat_inducing.data_ptr(), out.data_ptr(),
         _stream(out))
    return out


def stats_workspace(n, m, L, device):
    lib = _lib.load_library()
    return torch.empty(lib.svgp_stream_stats_workspace_elems(n, m, L), dtype=torch.float32, device=device)


def stats(K_nm, means, vars, ws=None, S=None, v=None, comm=None):
    """S (L, m, m) = K_nm^T diag(1/var_l) K_nm and v (L, m) = K_nm^T (mean_l / var_l).  With `comm` (engine.RcclComm)
    every rank passes its row shard and the two results are summed over ranks on the same stream (the
    all-reduce of (L, m, m) of config 5)."""
    _check(K_nm, "K_nm"); _check(means, "means"); _check(vars, "vars")
    n, m = K_nm.shape
    L = means.shape[1]
    assert means.shape == (n, L) and vars.shape == (n, L)
    dev = K_nm.device
    ws = stats_workspace(n, m, L, dev) if ws is None else ws
    S = torch.empty((L, m, m), dtype=torch.float32, device=dev) if S is None else S
    v = torch.empty((L, m), dtype=torch.float32, device=dev) if v is None else v
    call("svgp_stream_stats_f32", n, m, L, K_nm.data_ptr(), means.data_ptr(), vars.data_ptr(), S.data_ptr(),
         v.data_ptr(), ws.data_ptr(), ws.numel(), _stream(K_nm))
    if comm is not None:
        comm.all_reduce(S, _stream(K_nm))
        comm.all_reduce(v, _stream(K_nm))
    return S, v


def precompute_GP_params_f32(kd, means, vars, aux_data, inducing_index_points, table=None, K_mm=None):
    """precompute_GP_params_SVGPVAE (SVGPVAE_model.py:989-1023) for N-sized float32 inputs:
    returns (mean_terms (L, m), inv_Sigma_l (L, m, m)).  K_nm and the statistics are the float32 streaming
    kernels; Sigma_l = K_mm + S_l is inverted (no jitter, :1014) by the float64 batched SPD inverse."""
    fr = features(kd, aux_data, inducing=False, table=table)
    fi = features(kd, inducing_index_points, inducing=True)
    n, m = aux_data.shape[0], inducing_index_points.shape[0]
    K_nm = knm(kd, fr, n, fi, m)
    if K_mm is None:
        K_mm = knm(kd, fi, m, fi, m)
    S, v = stats(K_nm, means, vars)
    Sigma = (K_mm[None] + S).double().contiguous()
    L = Sigma.shape[0]
    lib = _lib.load_library()
    wsz = lib.svgp_spd_inverse_workspace_elems(m, L)
    w = torch.empty(max(int(wsz), 1), dtype=torch.float64, device=Sigma.device)
    logdet = torch.empty(L, dtype=torch.float64, device=Sigma.device)
    call("svgp_spd_inverse_batched", m, L, Sigma.data_ptr(), logdet.data_ptr(), w.data_ptr(), _stream(Sigma))
    inv = Sigma   # inverted in place
    vd = v.double().contiguous()
    mean_terms = torch.empty(L, m, dtype=torch.float64, device=Sigma.device)
    # mean_terms_l = Sigma_l^-1 v_l: the library's batched GEMM with one right-hand column per channel
    call("svgp_dgemm_batched", 0, 0, m, 1, m, 1.0, inv.data_ptr(), m, m * m, vd.data_ptr(), 1, m, 0.0, mean_terms.data_ptr(), 1, m,
         L, _stream(Sigma))
    return mean_terms.float(), inv.float()
