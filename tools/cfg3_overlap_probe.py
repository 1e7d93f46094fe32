"""Tuning aid: the two branches of the config-3 step (side: forward-factor tail + early reverse half; main: row stage, decoder
forward / reverse, reverse statistics) through the individual entry points, each alone and both together on two streams --
HIP events per branch.  What the overlap of the step can and does give."""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from svgp_vae_amd import _lib
from svgp_vae_amd.engine import MnistStepEngine
B, M_IND, MDIM = 1024, 256, 32
params, images, aux, eps = bench.synthetic_problem(0, B, M_IND, MDIM)
eng = MnistStepEngine(M_IND, bench.L, MDIM, bench.N_OBJ, N_train=bench.N_TRAIN, jitter=1e-6, clip_qs=True, geco=True, kappa_squared=0.020,
                      alpha=0.99, beta=0.001, lr=1e-3, b_max=B, device="cuda:0")
eng.load_params(params)
dev = eng.device
d_img, d_aux, d_eps = (torch.tensor(x, dtype=torch.float64, device=dev).contiguous() for x in (images, aux, eps))
eng.bind(d_img, d_aux, d_eps)
eng.run(adam=False); eng.synchronize()
cfg, ws, st = C.byref(eng.cfg), eng.ws.data_ptr(), eng.state.data_ptr()
th, im = eng.theta.data_ptr(), d_img.data_ptr()
main = eng.stream
side = torch.cuda.Stream(device=dev)
def side_branch(s):
    _lib.call("svgp_gp_factor_fwd_aji_tail", cfg, ws, s.cuda_stream)
    _lib.call("svgp_gp_factor_bwd_early", cfg, ws, st, s.cuda_stream)
def main_branch(s):
    _lib.call("svgp_gp_posterior_fwd", cfg, d_eps.data_ptr(), ws, st, s.cuda_stream)
    _lib.call("svgp_mnist_decoder_fwd", cfg, th, im, ws, s.cuda_stream)
    _lib.call("svgp_mnist_decoder_bwd", cfg, th, im, ws, st, s.cuda_stream)
    _lib.call("svgp_gp_stats_bwd", cfg, ws, st, s.cuda_stream)
def ev(): return torch.cuda.Event(enable_timing=True)
def run(mode, reps=30):
    tm, ts, tt = [], [], []
    for _ in range(reps):
        _lib.call("svgp_gp_factor_fwd_defer_aji", cfg, ws, main.cuda_stream)
        e0, e1, e2, e3 = ev(), ev(), ev(), ev()
        e0.record(main)
        if mode == "main": main_branch(main); e1.record(main); e2 = e1
        elif mode == "side": side_branch(main); e2.record(main); e1 = e2
        elif mode == "serial": side_branch(main); e2.record(main); main_branch(main); e1.record(main)
        else:
            side.wait_event(e0)
            if mode == "both_side_first": side_branch(side); e2.record(side); main_branch(main); e1.record(main)
            else: main_branch(main); e1.record(main); side_branch(side); e2.record(side)
            main.wait_event(e2)
        e3.record(main)
        torch.cuda.synchronize()
        tm.append(e0.elapsed_time(e1) * 1e3); ts.append(e0.elapsed_time(e2) * 1e3); tt.append(e0.elapsed_time(e3) * 1e3)
    med = lambda x: float(np.median(x[5:]))
    print(f"{mode:18s} main branch done at {med(tm):7.1f} us, side branch done at {med(ts):7.1f} us, joined at {med(tt):7.1f} us", flush=True)
for mode in ("main", "side", "serial", "both_side_first", "both_main_first"):
    run(mode)
