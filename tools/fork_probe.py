"""Development probe: cost / benefit of the opt-in side-stream branch (env SVGP_SIDE_STREAMS=1: kernel-matrix reverse pass
beside the encoder reverse pass) in whole-step graph replay, per-phase graph replay and the eager in-order form."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, torch
from svgp_vae_amd.engine import MnistStepEngine
params, images, aux, eps = bench.synthetic_problem(0)
dev = torch.device("cuda:0")
eng = MnistStepEngine(32, 16, 8, 400, geco=True, b_max=256)
eng.load_params(params)
t = lambda x: torch.tensor(x, dtype=torch.float64, device=dev).contiguous()
eng.bind(t(images), t(aux), t(eps)); eng.run(adam=False); eng.synchronize()

def timeit(fn, reps=300):
    for _ in range(20): fn()
    eng.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    eng.synchronize(); return (time.perf_counter() - t0) / reps * 1e6

for flag in ("0", "1"):
    os.environ["SVGP_SIDE_STREAMS"] = flag
    eng.capture("full", adam=False)
    eng.capture_phases("ph", adam=False)
    full = timeit(lambda: eng.replay("full"))
    ph = [timeit(lambda k=k: eng.replay(("ph", k))) for k in range(4)]
    eager = timeit(lambda: eng.run(adam=False), reps=100)
    print("SVGP_SIDE_STREAMS=" + flag, "full graph %.1f us" % full, "phase graphs", [round(x, 1) for x in ph], "eager %.1f" % eager, flush=True)
