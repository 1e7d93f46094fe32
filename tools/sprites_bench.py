"""Informational timing of the SPRITES step (BASELINE configs[3] shape, one GPU): b=500 frames (10 characters x 50),
L=64, L_action=8, L_character=16, m inducing points, jitter 0.01, cosine-normalised linear kernels, GECO."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from svgp_vae_amd import sprites as S
m = int(sys.argv[1]) if len(sys.argv) > 1 else 800
b, frames, L, La, Lc, n_act = 500, 50, 64, 8, 16, 72
rs = np.random.RandomState(0)
svgp = S.spritesSVGP(False, False, rs.normal(0, 1.5, (m, La + Lc)), 'main', 0.01, 50000, La, rs.normal(0, 1.5, (n_act, La)),
                     Lc, L, K_obj_normalize=True)
eng = S.SpritesStepEngine(S.spritesVAE(L), S.sprites_representation_network(Lc), svgp, b_max=b, seg_len=frames, geco=True,
                          kappa_squared=0.0075, clip_grad=1e6)
dev = eng.dev
img = torch.rand(b, 64, 64, 3, dtype=torch.float64, device=dev)
ids = torch.tensor(rs.randint(0, n_act, b), dtype=torch.float64, device=dev)
for _ in range(3):
    eng.step(img, ids, None, adam=True)
eng.stream.synchronize(); torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    eng.step(img, ids, None, adam=True)
eng.stream.synchronize(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
sc = eng.scalars()
print(json.dumps({"workload": f"SPRITES step b={b} L={L} m={m} float64", "ms_per_step": dt * 1e3, "steps_per_s": 1 / dt,
                  "elbo": sc["elbo"], "recon_loss": sc["recon_loss"], "finite": bool(np.isfinite(sc["elbo"]))}))
