"""Development probe (round 6): the decoder reverse pass as data half + weight half (riders of the reverse factor launch).
Prints (1) equality of the two forms, (2) stand-alone times of every piece, (3) the eager config-2 step with SVGP_DEC_SPLIT=0/1
(the env switch is read once per process, so the two steps are timed in child processes)."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))


def child():
    import torch
    import bench
    from svgp_vae_amd import _lib
    from svgp_vae_amd.engine import MnistStepEngine
    params, images, aux, eps = bench.synthetic_problem(0, 256, 32, 8)
    dev = torch.device("cuda:0")
    eng = MnistStepEngine(32, 16, 8, 400, geco=True, b_max=256)
    eng.load_params(params)
    t = lambda x: torch.tensor(x, dtype=torch.float64, device=dev).contiguous()
    eng.bind(t(images), t(aux), t(eps))
    for _ in range(30):
        eng.run(adam=True)
    eng.synchronize()
    reps = 400
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.run(adam=True)
        eng.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    print(json.dumps(dict(split=os.environ.get("SVGP_DEC_SPLIT", "1"), types=os.environ.get("SVGP_DEC_RIDER_TYPES", ""), merge=os.environ.get("SVGP_ENC_KM_MERGE", ""), sum_merge=os.environ.get("SVGP_SUM_MERGE", ""), stat_merge=os.environ.get("SVGP_STAT_MERGE", ""), aji_dec=os.environ.get("SVGP_AJI_DEC", ""), step_us=best, elbo=eng.scalars()["elbo"])), flush=True)


def main():
    import torch
    import bench
    from svgp_vae_amd import _lib
    from svgp_vae_amd.engine import MnistStepEngine
    lib = _lib.load_library()
    params, images, aux, eps = bench.synthetic_problem(0, 256, 32, 8)
    dev = torch.device("cuda:0")
    eng = MnistStepEngine(32, 16, 8, 400, geco=True, b_max=256)
    eng.load_params(params)
    t = lambda x: torch.tensor(x, dtype=torch.float64, device=dev).contiguous()
    eng.bind(t(images), t(aux), t(eps))
    eng.run(adam=False)
    eng.synchronize()
    cfg, th, ws, st = C.byref(eng.cfg), eng.theta.data_ptr(), eng.ws.data_ptr(), eng.state.data_ptr()
    img = eng._bound[0].data_ptr()
    s = eng.stream.cuda_stream
    n_dec = eng.pl.n_vae - eng.pl.n_enc

    def parts():
        eng.synchronize()
        return eng.ws_view("part_dec", (eng.wl.n_part, n_dec)).clone(), eng.ws_view("zbar", (256, 16)).clone()

    _lib.call("svgp_mnist_decoder_bwd", cfg, th, img, ws, st, s)
    p0, z0 = parts()
    eng.ws_view("part_dec", (eng.wl.n_part, n_dec)).zero_(); eng.ws_view("zbar", (256, 16)).zero_()
    _lib.call("svgp_mnist_decoder_bwd_data", cfg, th, img, ws, st, s)
    for thr, nty in ((256, 1), (512, 1), (256, 2), (256, 3), (512, 3)):
        eng.ws_view("part_dec", (eng.wl.n_part, n_dec)).zero_()
        _lib.call("svgp_mnist_decoder_bwd_weights", cfg, img, ws, st, thr, nty, s)
        p1, z1 = parts()
        g0, g1 = p0.sum(0), p1.sum(0)
        print(f"threads {thr} x {nty}: zbar max abs diff {float((z0 - z1).abs().max()):.3e}; summed partials rel diff "
              f"{float((g0 - g1).abs().max() / g0.abs().max()):.3e}; per-partial max rel {float((p0 - p1).abs().max() / p0.abs().max()):.3e}", flush=True)
    eng.ws_view("part_dec", (eng.wl.n_part, n_dec)).zero_()
    _lib.call("svgp_gp_factor_bwd_nofinal_wgrad", cfg, img, ws, st, s)
    p2, _ = parts()
    print(f"riders: per-partial max rel {float((p0 - p2).abs().max() / p0.abs().max()):.3e}", flush=True)

    e0, e1 = C.c_void_p(), C.c_void_p()
    _lib.call("svgp_event_create", C.byref(e0)); _lib.call("svgp_event_create", C.byref(e1))

    def timeit(sym, args, reps=200):
        fn = getattr(lib, sym)
        for _ in range(5):
            _lib.check(fn(*args))
        best = 1e9
        for _ in range(3):
            _lib.call("svgp_event_record", e0, s)
            for _ in range(reps):
                _lib.check(fn(*args))
            _lib.call("svgp_event_record", e1, s)
            ms = C.c_float()
            _lib.call("svgp_event_elapsed_ms", e0, e1, C.byref(ms))
            best = min(best, ms.value * 1e3 / reps)
        return best

    rows = [("decoder_bwd (one kernel)", "svgp_mnist_decoder_bwd", (cfg, th, img, ws, st, s)),
            ("decoder_bwd_data", "svgp_mnist_decoder_bwd_data", (cfg, th, img, ws, st, s)),
            ("decoder_bwd_data_pre", "svgp_mnist_decoder_bwd_data_pre", (cfg, th, img, ws, st, s)),
            ("decoder_bwd_data_pre_aji", "svgp_mnist_decoder_bwd_data_pre_aji", (cfg, th, img, ws, st, s)),
            ("decoder_fwd", "svgp_mnist_decoder_fwd", (cfg, th, img, ws, s)),
            ("decoder_fwd_pre", "svgp_mnist_decoder_fwd_pre", (cfg, th, img, ws, s)),
            ("encoder_kernel_matrix_fwd", "svgp_mnist_encoder_kernel_matrix_fwd", (cfg, th, img, eng._bound[1].data_ptr(), ws, s)),
            ("decoder_bwd_weights 256 x 1", "svgp_mnist_decoder_bwd_weights", (cfg, img, ws, st, 256, 1, s)),
            ("decoder_bwd_weights 512 x 1", "svgp_mnist_decoder_bwd_weights", (cfg, img, ws, st, 512, 1, s)),
            ("decoder_bwd_weights 256 x 2", "svgp_mnist_decoder_bwd_weights", (cfg, img, ws, st, 256, 2, s)),
            ("decoder_bwd_weights 256 x 3", "svgp_mnist_decoder_bwd_weights", (cfg, img, ws, st, 256, 3, s)),
            ("decoder_bwd_weights 512 x 3", "svgp_mnist_decoder_bwd_weights", (cfg, img, ws, st, 512, 3, s)),
            ("gp_factor_bwd_nofinal", "svgp_gp_factor_bwd_nofinal", (cfg, ws, st, s)),
            ("gp_factor_bwd_nofinal_wgrad", "svgp_gp_factor_bwd_nofinal_wgrad", (cfg, img, ws, st, s)),
            ("gp_stats_bwd", "svgp_gp_stats_bwd", (cfg, ws, st, s)),
            ("gp_stats_factor_bwd_wgrad", "svgp_gp_stats_factor_bwd_wgrad", (cfg, img, ws, st, s)),
            ("kernel_matrix_bwd_partials", "svgp_kernel_matrix_bwd_partials", (cfg, th, eng._bound[1].data_ptr(), ws, s)),
            ("encoder_bwd", "svgp_mnist_encoder_bwd", (cfg, th, img, ws, s)),
            ("encoder_bwd_km", "svgp_mnist_encoder_bwd_km", (cfg, th, img, eng._bound[1].data_ptr(), ws, s)),
            ("encoder_bwd_km_sum", "svgp_mnist_encoder_bwd_km_sum", (cfg, th, img, eng._bound[1].data_ptr(), ws, st, s)),
            ("gp_posterior_bwd_with_final", "svgp_gp_posterior_bwd_with_final", (cfg, ws, st, s)),
            ("gp_posterior_bwd_rows", "svgp_gp_posterior_bwd_rows", (cfg, ws, st, s)),
            ("grad_reduce_all", "svgp_mnist_grad_reduce_all", (cfg, eng._bound[1].data_ptr(), ws, s))]
    for name, sym, args in rows:
        print(f"{name:32s} {timeit(sym, args):7.2f} us", flush=True)
    for flag, sm, st_, aj in (("0", "0", "0", "0"), ("1", "1", "0", "0"), ("1", "1", "1", "0"), ("1", "1", "1", "1"), ("1", "1", "1", "0"),
                              ("1", "1", "1", "1"), ("1", "1", "1", "0"), ("1", "1", "1", "1")):
        env = dict(os.environ, SVGP_DEC_SPLIT=flag, SVGP_ENC_KM_MERGE=flag, SVGP_SUM_MERGE=sm, SVGP_STAT_MERGE=st_, SVGP_AJI_DEC=aj)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:], flush=True)

if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
