"""Timing ablation helper (development tool): env SVGP_DBG_STOP selects early exits / variants."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, torch
from svgp_vae_amd.engine import MnistStepEngine
params, images, aux, eps = bench.synthetic_problem(0)
dev = torch.device("cuda:0")
eng = MnistStepEngine(32, 16, 8, 400, geco=True, b_max=256)
eng.load_params(params)
t = lambda x: torch.tensor(x, dtype=torch.float64, device=dev).contiguous()
eng.bind(t(images), t(aux), t(eps)); eng.run(adam=False); eng.synchronize()
stops = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3, 4, 0]
keys = sys.argv[2].split(",") if len(sys.argv) > 2 else ["gp_factor_fwd"]
for stop in stops:
    os.environ["SVGP_DBG_STOP"] = str(stop)
    rows = {r["stage"]: r["us"] for r in bench.time_stages(eng, reps=100)}
    print("stop", stop, {k: round(rows[k], 1) for k in keys}, flush=True)
os.environ["SVGP_DBG_STOP"] = "0"
