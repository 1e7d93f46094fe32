"""The float32 question of VERDICT r4 item 4: with SE x SE kernels (--K_SE) K_mm is full rank, so the 1 / jitter-sized cancellations
that sink float32 GP products under the rank-deficient linear kernel are absent.  For cfg.gemm_f32 = 0 / 2 / 1 (GP products in
float64 / the statistics products in float32 / every product in float32; storage float64, networks float32) at the benchmarked
500-frame, m = 800 size: every gradient against the oracle (float64 autograd) next to the tolerance of
tests/test_gpu_fullsize.py (max(2e-5, 20 x the oracle's response to a one-float32-ulp input perturbation)), and the step time.
Information only (prints a table + one JSON line); run on the GPU box:  python tools/sprites_gemm_f32_probe.py [linear|se]"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

from tests import helpers as H
from tests import test_gpu_fullsize as T

DT = torch.float64
K_SE = (sys.argv[1] if len(sys.argv) > 1 else "se") == "se"
from svgp_vae_amd import sprites as S

c = T._sprites500_case(K_SE)
b, frames, L, La, Lc, n_act, m = c["dims"]
params, gp, images, ids, eps, want, wgrads = c["params"], c["gp"], c["images"], c["ids"], c["eps"], c["want"], c["wgrads"]
SEK = ("l_action", "sigma_action", "l_character", "sigma_character")
pert = c["p32"][1]
out = {}
for gf in (0, 2, 1):
    svgp = S.spritesSVGP(False, False, gp["inducing_index_points"].numpy(), 'main', c["jitter"], c["N_train"], La,
                         gp["GPLVM_action"].numpy(), Lc, L, fixed_GP_params=False, fixed_GPLVM=False, K_obj_normalize=True, K_SE=K_SE)
    init = dict(params)
    init["se"] = torch.stack([gp[k] for k in SEK])
    eng = S.SpritesStepEngine(S.spritesVAE(L), S.sprites_representation_network(Lc), svgp, b_max=b, seg_len=frames, clip_qs=False,
                              geco=True, kappa_squared=0.0075, beta=0.001, clip_grad=1e6, params=init, net_dtype=torch.float32,
                              gemm_f32=gf)
    eng.set_scalars(c_ma=0.0, lagrange=1.0, alpha=0.0)
    dev = eng.dev
    di, dd, de = images.to(dev, eng.ndt), ids.to(dev, DT), eps.to(dev)
    eng.step(di, dd, de, adam=False)
    got = eng.outputs()
    gr = eng.grads
    rows, worst = [], 0.0
    for k, w in list(wgrads.items()) + ([("se", torch.stack([wgrads[k] for k in SEK]))] if K_SE else []):
        if k in SEK:
            continue
        pk = torch.stack([pert[j] for j in SEK]) if k == "se" else pert[k]
        err, tol = H.relerr(gr[k], w), max(2e-5, 20.0 * H.relerr(pk, w))
        rows.append((k, err, tol))
        worst = max(worst, err / tol)
    elbo_rel = abs(float(got[0]) - float(want[0])) / abs(float(want[0]))
    for _ in range(2):
        eng.step(di, dd, None, adam=False)
    eng.stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.step(di, dd, None, adam=False)
    eng.stream.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"--- kernel {'SE x SE' if K_SE else 'linear'}  gemm_f32 = {gf}: step {ms:.2f} ms, ELBO rel {elbo_rel:.2e}, worst gradient err / tol = {worst:.2f}")
    for k, err, tol in sorted(rows, key=lambda r: -r[1] / r[2])[:6]:
        print(f"      {k:28s} rel {err:.2e}  tol {tol:.2e}")
    out[f"gemm_f32_{gf}"] = dict(step_ms=round(ms, 3), elbo_rel=elbo_rel, worst_err_over_tol=round(worst, 3),
                                 finite=bool(all(torch.isfinite(v).all() for v in gr.values())),
                                 grads={k: dict(rel=err, tol=tol) for k, err, tol in rows})
    del eng
print(json.dumps(dict(kernel="se" if K_SE else "linear", m=m, frames=b, results=out)))
