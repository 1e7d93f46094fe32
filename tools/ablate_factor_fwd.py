import os, sys, json, subprocess
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, torch, ctypes as C
from svgp_vae_amd.engine import MnistStepEngine
params, images, aux, eps = bench.synthetic_problem(0)
dev = torch.device("cuda:0")
eng = MnistStepEngine(32, 16, 8, 400, geco=True, b_max=256)
eng.load_params(params)
t = lambda x: torch.tensor(x, dtype=torch.float64, device=dev).contiguous()
eng.bind(t(images), t(aux), t(eps)); eng.run(adam=False); eng.synchronize()
for stop in (1, 2, 3, 4, 5, 6, 7, 0):
    os.environ["SVGP_DBG_STOP"] = str(stop)
    rows = {r["stage"]: r["us"] for r in bench.time_stages(eng, reps=100)}
    print("stop", stop, {k: round(rows[k], 1) for k in ("gp_factor_fwd", "decoder_bwd")}, flush=True)
os.environ["SVGP_DBG_STOP"] = "0"
