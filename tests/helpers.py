"""Shared test helpers: toy problems and the oracle's stage-by-stage intermediates."""
import math

import numpy as np
import torch

from oracle import staged_gp as SG
from oracle import svgpvae_oracle as O

DT = torch.float64


def toy_problem(b=40, m=12, L=3, M=4, n_obj=20, seed=0, with_table=True):
    """Random rotated-MNIST-shaped problem.  Returns (params dict of float64 CPU tensors, images, aux, eps)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, dtype=DT, generator=g)
    u = lambda *s: torch.rand(*s, dtype=DT, generator=g)
    params = {k: torch.tensor(v, dtype=DT) for k, v in O.glorot_uniform_init(L, seed).items()}
    for k in list(params):
        if k.endswith("_b"):
            params[k] = 0.1 * r(*params[k].shape)
    params["inducing_index_points"] = torch.cat([torch.arange(m, dtype=DT)[:, None], u(m, 1) * 6.28,
                                                 r(m, M) * 1.5], 1)
    params["l_GP"] = torch.tensor(1.2, dtype=DT)
    params["amplitude"] = torch.tensor(0.9, dtype=DT)
    if with_table:
        params["object_vectors"] = r(n_obj, M) * 1.5
    ids = torch.randint(0, max(n_obj, 1), (b, 1), generator=g).to(DT)
    aux = torch.cat([ids, u(b, 1) * 6.28, r(b, M) * 1.5], 1).contiguous()
    images = torch.clamp(0.142 + 0.316 * r(b, 28, 28, 1), -0.2, 1.2).contiguous()
    eps = r(b, L).contiguous()
    return params, images, aux, eps


def golden_problem(golden, rows=slice(0, 256)):
    gin, _ = golden
    params = {k[4:]: torch.tensor(v, dtype=DT) for k, v in gin.items() if k.startswith("vae_")}
    for k in ("inducing_index_points", "l_GP", "amplitude", "object_vectors"):
        params[k] = torch.tensor(gin[k], dtype=DT)
    images, aux, eps = (torch.tensor(gin[k][rows], dtype=DT).contiguous() for k in ("images", "aux", "epsilon"))
    return params, images, aux, eps


def oracle_stages(params, images, aux, eps, *, N_train, jitter, clip_qs, geco, beta, lagrange_mult=1.0,
                  K_obj_normalize=False, b_global=None):
    """Every intermediate the HIP workspace exposes, from the oracle (single rank)."""
    L = eps.shape[1]
    b = images.shape[0]
    bg = float(b if b_global is None else b_global)
    c = N_train / bg
    vae = O.MnistVAE(params, L)
    mu, var_raw = vae.encode(images)
    var = O.clip_by_value(var_raw, 1e-3, 10.0) if clip_qs else var_raw
    ov = params.get("object_vectors")
    K, Kn, knn = SG.kernel_matrix_fwd(aux, params["inducing_index_points"], ov, params["l_GP"],
                                      params["amplitude"], K_obj_normalize)
    p = O.reciprocal_no_nan(var)
    S, v, _ = SG.gp_stats(Kn, p, p * mu)
    f = SG.gp_factor_fwd(K, S, v, jitter, c)
    ps = SG.gp_posterior_fwd(Kn, knn, mu, var, eps, f, c)
    recon = vae.decode(ps["z"])
    out = dict(qnet_mu=mu, qnet_var_raw=var_raw, qnet_var=var, K=K, Kn=Kn, knn=knn, S=S, v=v, Ki=f["Ki"],
               ldK=f["ldK"].reshape(1), Si=f["Si"], t=f["t"], G=f["G"], A=f["A"], Aji=f["Aji"], mu_hat=f["mu"],
               u=f["u"], KL=f["KL"], q=ps["q"], p_m=ps["p_m"], p_v=ps["p_v"], e=ps["e"], d=ps["d"], z=ps["z"],
               recon=recon.reshape(b, 784),
               M2=f["Ki"][None] @ f["A"] @ f["Ki"][None])
    return out


def engine_for(params, b, *, geco, clip_qs=True, N_train=4050.0, jitter=1e-6, beta=0.001, lr=1e-3,
               K_obj_normalize=False, alpha=0.99, kappa_squared=0.020, **kw):
    from svgp_vae_amd.engine import MnistStepEngine
    m, Mp2 = params["inducing_index_points"].shape
    L = params["dec_d_w"].shape[0]
    n_obj = params["object_vectors"].shape[0] if "object_vectors" in params else 0
    eng = MnistStepEngine(m, L, Mp2 - 2, n_obj, N_train=N_train, jitter=jitter, clip_qs=clip_qs, geco=geco,
                          K_obj_normalize=K_obj_normalize, beta=beta, lr=lr, alpha=alpha,
                          kappa_squared=kappa_squared, b_max=b, **kw)
    eng.load_params(params)
    return eng


def relerr(a, b):
    a = torch.as_tensor(a, dtype=DT).cpu().reshape(-1)
    b = torch.as_tensor(b, dtype=DT).cpu().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))
