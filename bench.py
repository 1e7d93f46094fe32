"""SVGPVAE_Hensman train steps/sec on MI355X (BASELINE.json metric) and the roofline-relevant side workloads.

    python bench.py --gpus 1 --steps 200 --warmup 20                      # BASELINE configs[1] (the metric)
    python bench.py --workload cfg3|sprites800|cfg5                       # BASELINE configs[2] / [3] / [4]
    python bench.py --gpus N --steps K --warmup W                         # N > 1: spawns the launcher below as a child
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Default workload (cfg2): one "step" = one full training step (encoder, kernel matrices, sparse-GP block, decoder,
reverse pass, TF1 Adam, GECO state update) on one 256-row rotated-MNIST-shaped batch per GPU (BASELINE configs[1]:
m=32 inducing points, L=16, GPLVM dim 8, N_train=4050, float64 like the reference).  Weak scaling: every rank keeps
256 rows, the global batch is 256*N rows coupled through the sufficient-statistics all-reduces; `value` =
N * steps / time = 256-row batches trained per second.  With N > 1 a `"scaling": "strong"` line (the 256-row global batch of
the headline configuration split over the ranks; value = steps / time) is printed BEFORE the weak line, which stays last.
Inputs are synthetic, generated once and resident in HBM before the timed region.

Timing: after W warm-up steps, R (`--repeats`, default 5) blocks of EXACTLY K steps, each bracketed by barrier +
synchronize on both sides and reduced with MAX over ranks; the reported ms_per_step / value come from the MEDIAN block
(a 20-step block is a 4 ms sample; one scheduling hiccup would move a single block by 10 %).

Every workload prints ONE JSON line with `roofline` (dominant stage, HIP events on the launch stream, algorithmic
work model of SURVEY 8d / DESIGN section 5) and `cpu_baseline` (the oracle timed on this host's cores; N=1 only).
"""
import argparse
import ctypes as C
import glob
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

L, N_OBJ, N_TRAIN = 16, 400, 4050.0
F64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak (AMD datasheet; = 1/2 of the 157.3 TF f32 rate
                           # listed in MI355X_MICROARCH.md, which has no f64 row)
F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
HBM_PEAK_GBS = 8000.0


# =====================================================================================================================
# shared helpers
# =====================================================================================================================
def dist_env():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def timed_blocks(step, sync, steps, warmup, repeats, multi, dev):
    """W warm-up steps, then `repeats` blocks of exactly `steps` steps; per block barrier + synchronize on both sides,
    MAX over ranks.  Returns the list of block times in seconds (identical on every rank)."""
    import torch.distributed as dist
    for _ in range(warmup):
        step()
    out = []
    for _ in range(repeats):
        sync()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        el = time.perf_counter() - t0
        if multi:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        out.append(el)
    return out


def roofline_of(flops, nbytes, us, peak_tflops, **extra):
    """Roofline object of one launch: bound chosen by the arithmetic intensity of the ALGORITHMIC work."""
    ai = flops / max(nbytes, 1.0)
    if ai >= peak_tflops * 1e12 / (HBM_PEAK_GBS * 1e9):
        ach = flops / (us * 1e-6) / 1e12
        r = {"bound": "mfma", "achieved": ach, "peak": peak_tflops, "unit": "TFLOP/s", "frac": ach / peak_tflops}
    else:
        ach = nbytes / (us * 1e-6) / 1e9
        r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
    r.update(traffic=None, launch_us=us, algorithmic_flops=float(flops), algorithmic_bytes=float(nbytes))
    r.update(extra)
    return r


def _traffic_src(path):
    """`profiles/<tag>_pmc_traffic.json @ <commit the counters were collected at>` (profiles/<tag>_commit.txt)."""
    rel = os.path.relpath(path, ROOT)
    try:
        commit = open(path.replace("_pmc_traffic.json", "_commit.txt")).read().strip()
        return f"{rel} @ {commit}"
    except Exception:
        return rel


def committed_traffic(kernel_key, per_pass_of=None):
    """HBM bytes per launch of `kernel_key` from the newest committed rocprofv3 PMC summary (profiles/*_pmc_traffic.json,
    collected with tools/pmc_summary.py in separate --pmc passes).  bench.py cannot run the profiler on itself, so this
    is an EARLIER measurement: the source file is reported next to the value and it is null when none matches."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            ks = json.load(open(path))["kernels"]
            for key in (kernel_key, "cfg5:" + kernel_key, "cfg3:" + kernel_key, "sp800:" + kernel_key):
                if key in ks:
                    k = ks[key]
                    if per_pass_of and (key.split(":")[0] + ":" + per_pass_of) in ks and "hbm_bytes_total_corrected" in k:
                        # several launches (instantiations) of the kernel per pass: bytes of all of them / passes profiled
                        return k["hbm_bytes_total_corrected"] / ks[key.split(":")[0] + ":" + per_pass_of]["dispatches"], _traffic_src(path)
                    return k["hbm_bytes_per_launch_corrected"], _traffic_src(path)
        except Exception:
            pass
    return None, None


def committed_step_traffic(workload):
    """HBM bytes of ONE optimiser step of `workload` (cfg2 / cfg3 / sp800) from the newest committed PMC summary that has it."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            w = json.load(open(path)).get("workloads", {}).get(workload)
            if w:
                return w["hbm_bytes_per_step_corrected"], _traffic_src(path)
        except Exception:
            pass
    return None, None


def cpu_worker_call(kind, threads, budget_s, extra=()):
    """Runs one CPU-baseline leg in a CHILD process whose OpenMP / MKL thread count is fixed by the environment before
    torch is imported.  (torch.set_num_threads after the pools exist is not safe with this build: on the GPU box the
    batched LU behind torch.linalg.inv returned corrupt pivots after a thread-count change, in the build container the
    same call dead-locked.)  The child regenerates the synthetic problem from its seeds and prints one JSON object."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), CUDA_VISIBLE_DEVICES="",
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", kind, "--budget", str(budget_s), *extra],
                       env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError(f"cpu baseline worker {kind} failed:\n{r.stderr[-2000:]}")
    return json.loads(r.stdout.strip().splitlines()[-1])


def host_threads(cap):
    return max(1, min(cap, os.cpu_count() or 8))


def emit(line):
    # every line: the minimum next to the median of the timed blocks, and the blocks that took > 3x the median (a fresh box's
    # first blocks can include a clock ramp or a paged-in library: BENCH_r02 had one 77 ms block among 3.9 ms ones)
    if "block_ms" in line:
        b = line["block_ms"]
        med = float(np.median(b))
        line["block_ms_min"], line["block_ms_median"] = float(min(b)), med
        line["block_outliers"] = [{"index": i, "ms": x, "x_median": round(x / med, 1)} for i, x in enumerate(b) if x > 3 * med]
    # RCCL prints a version banner through C stdio at communicator creation; flush it first so the JSON line is the
    # last line on stdout
    C.CDLL(None).fflush(None)
    sys.stdout.flush()
    print(json.dumps(line), flush=True)


# =====================================================================================================================
# cfg2 / cfg3: rotated-MNIST SVGPVAE_Hensman step
# =====================================================================================================================
def synthetic_problem(rank, B, M_IND, MDIM, seed=0):
    """SURVEY 8d config 2/3 synthetic inputs; parameters identical on all ranks, data differs per rank."""
    from svgp_vae_amd.VAE_utils import glorot_uniform_params
    rs = np.random.RandomState(seed)
    params = dict(glorot_uniform_params(L, seed))
    angles16 = np.linspace(0, 2 * np.pi, 17)[:-1]
    ip = np.concatenate([np.repeat(angles16, M_IND // 16)[:, None], rs.normal(0, 1.5, (M_IND, MDIM))], 1)
    params["inducing_index_points"] = np.concatenate([np.arange(M_IND)[:, None].astype(float), ip], 1)
    params["l_GP"] = np.array(1.0)
    params["amplitude"] = np.array(1.0)
    params["object_vectors"] = rs.normal(0, 1.5, (N_OBJ, MDIM))
    rd = np.random.RandomState(1000 + rank)
    train_angles = np.delete(angles16, 7)
    ids = rd.randint(0, N_OBJ, B).astype(float)
    aux = np.concatenate([ids[:, None], rd.choice(train_angles, B)[:, None],
                          params["object_vectors"][ids.astype(int)]], 1)
    images = np.clip(rd.normal(0.142, 0.316, (B, 28, 28, 1)), -0.2, 1.2)
    eps = rd.randn(B, L)
    return params, images, aux, eps


def library_comm(multi, local_rank, dev, timeout_s=180.0, make_id=None, make_comm=None):
    """The library's own RCCL communicator, decided collectively: svgp_vae_amd.dp.library_comm (shared with the *_experiment.py
    drivers).  Test hooks: SVGP_BENCH_FAIL_LIBCOMM=1 (every rank), SVGP_BENCH_FAIL_LIBCOMM_RANK=<r>[:id|:init] (one rank)."""
    from svgp_vae_amd.dp import library_comm as impl
    return impl(multi, local_rank, dev, timeout_s=timeout_s, make_id=make_id, make_comm=make_comm)


def stage_table(eng, B, M_IND):
    """(name, C symbol, args, algorithmic flops, algorithmic bytes, implementation-partials bytes) per stage and GPU.
    Flop/byte models: DESIGN.md section 5.  ALGORITHMIC bytes = inputs read once + outputs written once; the
    per-workgroup weight-gradient partials the two conv reverse kernels write (and the reduction re-reads) are
    implementation traffic and are listed separately."""
    cfg, th, ws, st = C.byref(eng.cfg), eng.theta.data_ptr(), eng.ws.data_ptr(), eng.state.data_ptr()
    img, aux, eps = (t.data_ptr() if t is not None else None for t in eng._bound)
    s = eng.stream.cuda_stream
    b, m, Lc = B, M_IND, L
    D = eng.base["M"] + 1
    enc_mac, dec_mac = 35_592, 207_872            # MACs per image (SURVEY App. B shapes)
    act_enc, act_dec = 1352 + 288 + 32, 128 + 512 + 1568
    n_enc, n_dec = eng.pl.n_enc, eng.pl.n_vae - eng.pl.n_enc
    n_part = eng.wl.n_part
    f8 = 8.0
    return [
        ("encoder_kernel_matrix_fwd", "svgp_mnist_encoder_kernel_matrix_fwd", (cfg, th, img, aux, ws, s),
         2 * enc_mac * b + (b * m + m * m) * (2 * D + 12),
         f8 * (b * (784 + act_enc + 3 * Lc) + b * m + m * m + b + b * (D + 1) + n_enc), 0.0),
        ("gp_stats_fwd", "svgp_gp_stats_fwd", (cfg, ws, s), 3 * Lc * b * m * m + 2 * m ** 3,
         f8 * (b * m + 2 * b * Lc + Lc * m * (m + 1)), 0.0),
        # m > 64 (round 4, "W form"): per channel the inverse + G = Si K + A = K G; shared: Kn Ki, W = (Kn Ki) K, K Ki.
        # m <= 64: the LDS-resident kernel still forms T = A Ki and M2 = Ki A Ki (4 products per channel)
        ("gp_factor_fwd", "svgp_gp_factor_fwd_defer_aji", (cfg, ws, s),
         (Lc * (2 * m ** 3 + 2 * 2 * m ** 3) + 4 * b * m * m + 2 * m ** 3) if m > 64 else
         (Lc * (2 * (m ** 3 / 3 + 2 * m ** 3 / 3) + 4 * 2 * m ** 3) + 2 * b * m * m), f8 * Lc * (5 if m > 64 else 7) * m * m, 0.0),
    ] + ([("gp_factor_fwd_aji_tail (side stream in the step)", "svgp_gp_factor_fwd_aji_tail", (cfg, ws, s), Lc * 2 * m ** 3,
           f8 * Lc * 2 * m * m, 0.0)] if m > 64 else []) + [
        # m <= 32 (round 6): without the deferred (A_hat + jI)^-1, which rides in the decoder's data-reverse launch below
        ("gp_posterior_fwd", "svgp_gp_posterior_fwd" if m <= 32 else "svgp_gp_posterior_fwd_with_aji", (cfg, eps, ws, st, s),
         4 * Lc * b * m * m, f8 * (Lc * 2 * m * m + b * m + 8 * b * Lc), 0.0),
        ("decoder_fwd", "svgp_mnist_decoder_fwd_pre" if m <= 64 else "svgp_mnist_decoder_fwd", (cfg, th, img, ws, s), 2 * dec_mac * b,
         f8 * (b * (Lc + act_dec + 2 * 784) + n_dec), 0.0),
        # m <= 64 (round 6): the data half alone (the chain to zbar; it also stores the pre-activation gradients d2, d1, dh0 for the
        # weight half, which rides in the reverse factor launch below); m > 64: the one-kernel form
    ] + ([("decoder_bwd_data", "svgp_mnist_decoder_bwd_data_pre_aji" if m <= 32 else "svgp_mnist_decoder_bwd_data_pre",
           (cfg, th, img, ws, st, s), 2 * dec_mac * b + (Lc * 2 * m ** 3 if m <= 32 else 0),
           f8 * (b * (512 + 1568 + 2 * 784 + Lc) + n_dec + (2 * Lc * m * m if m <= 32 else 0)), f8 * b * (1568 + 512 + 128))]
         if m <= 64 else
         [("decoder_bwd", "svgp_mnist_decoder_bwd", (cfg, th, img, ws, st, s), 4 * dec_mac * b,
           f8 * (b * (Lc + act_dec + 2 * 784 + Lc) + 2 * n_dec), f8 * n_part * n_dec)]) + [
        # m > 64: + the rank-local row terms [Qs; Pbar^T] = X^T Kn and (all rows local, b < 3 m) the statistic SW = W^T diag(p) W
    ] + ([("gp_stats_bwd", "svgp_gp_stats_bwd", (cfg, ws, st, s),
           3 * Lc * b * m * m + 4 * b * m * m + (Lc * b * m * m if b < 3 * m else 0),
           f8 * (b * m + 9 * b * Lc + Lc * m * (m + 2)), 0.0)] if m > 64 else []) + ([  # early: H = G (Ki - Aji), H G^T (+ SW = P^T S P when it is not formed over the rows); late: Si X, (Si X) Si + single matrices
          ("gp_factor_bwd_early (side stream in the step)", "svgp_gp_factor_bwd_early", (cfg, ws, st, s),
           Lc * (2 * m ** 3 + m ** 3 + (0 if b < 3 * m else 3 * m ** 3)), f8 * Lc * 6 * m * m, 0.0),
          ("gp_factor_bwd", "svgp_gp_factor_bwd_late", (cfg, ws, st, s), Lc * 2 * 2 * m ** 3 + 4 * 2 * m ** 3, f8 * Lc * 8 * m * m, 0.0)]
         if m > 64 else
         # ONE launch (single-GPU step, round 6): the reverse statistics (P L workgroups at the head; channel l waits for its own P) +
         # the reverse factor stage + the decoder's weight gradients as rider workgroups (3 per image)
         [("gp_stats_factor_bwd", "svgp_gp_stats_factor_bwd_wgrad", (cfg, img, ws, st, s),
           3 * Lc * b * m * m + Lc * 9 * 2 * m ** 3 + 2 * dec_mac * b,
           f8 * (b * m + 9 * b * Lc + Lc * m * (m + 2) + Lc * 14 * m * m + b * (Lc + act_dec + 2 * 784) + n_dec),
           f8 * (n_part * n_dec + b * (1568 + 512 + 128)))]) + [
        # m > 64: ONE (b, m, m) product per channel (Kn Ssym; Kn Si is the forward pass's) + Wbar P^T; m <= 64: three
        # m <= 64 (round 6): pass 1 alone (the per-channel row terms); pass 2 (the sums over channels) rides in the next launch
        ("gp_posterior_bwd", "svgp_gp_posterior_bwd_with_final" if m > 64 else "svgp_gp_posterior_bwd_rows", (cfg, ws, st, s),
         (2 * Lc * b * m * m + 2 * b * m * m) if m > 64 else 6 * Lc * b * m * m,
         f8 * (Lc * 3 * m * m + 2 * Lc * b * m + 12 * b * Lc), 0.0),
    ] + ([  # m <= 64 (round 6): pass 2 of the reverse row stage, then the kernel-matrix VJP workgroups (waiting for it), ride FIRST in
            # the encoder's reverse launch
        ("encoder_bwd_km", "svgp_mnist_encoder_bwd_km_sum", (cfg, th, img, aux, ws, st, s),
         4 * enc_mac * b + (2 * b * m + 2 * m * m) * (2 * D + 20),
         f8 * (b * (784 + act_enc + 3 * Lc) + 2 * n_enc + 2 * b * m + 2 * m * m + b * (D + 1) + N_OBJ * (D - 1)), f8 * n_part * n_enc)]
         if m <= 64 else [
        ("kernel_matrix_bwd", "svgp_kernel_matrix_bwd_partials", (cfg, th, aux, ws, s),
         (2 * b * m + 2 * m * m) * (2 * D + 20), f8 * (2 * b * m + 2 * m * m + b * (D + 1) + N_OBJ * (D - 1)), 0.0),
        ("encoder_bwd", "svgp_mnist_encoder_bwd", (cfg, th, img, ws, s), 4 * enc_mac * b,
         f8 * (b * (784 + act_enc + 3 * Lc) + 2 * n_enc), f8 * n_part * n_enc)]) + [
        ("grad_reduce", "svgp_mnist_grad_reduce_all", (cfg, aux, ws, s), n_part * (n_enc + n_dec),
         f8 * (n_enc + n_dec), f8 * n_part * (n_enc + n_dec)),
    ]


def time_stages(eng, B, M_IND, reps=50):
    """HIP events on the engine's own stream around `reps` back-to-back launches of each stage
    (inputs of every stage are valid after one full step).  Returns list of dicts sorted by time."""
    from svgp_vae_amd import _lib
    lib = _lib.load_library()
    s = eng.stream.cuda_stream
    e0, e1 = C.c_void_p(), C.c_void_p()
    _lib.call("svgp_event_create", C.byref(e0)); _lib.call("svgp_event_create", C.byref(e1))
    rows = []
    for name, sym, args, flops, nbytes, pbytes in stage_table(eng, B, M_IND):
        fn = getattr(lib, sym)
        for _ in range(3):
            _lib.check(fn(*args))
        _lib.call("svgp_event_record", e0, s)
        for _ in range(reps):
            _lib.check(fn(*args))
        _lib.call("svgp_event_record", e1, s)
        ms = C.c_float()
        _lib.call("svgp_event_elapsed_ms", e0, e1, C.byref(ms))
        rows.append(dict(stage=name, us=ms.value * 1e3 / reps, flops=float(flops), bytes=float(nbytes),
                         partials_bytes=float(pbytes)))
    _lib.call("svgp_event_destroy", e0); _lib.call("svgp_event_destroy", e1)
    return sorted(rows, key=lambda r: -r["us"])


def oracle_elbo(params, images, aux, eps):
    """ELBO of the explicit-eps GECO step from the oracle's efficient formulation (float64 torch-CPU)."""
    from oracle import svgpvae_oracle as O
    p = {k: torch.tensor(np.asarray(v), dtype=O.DT) for k, v in params.items()}
    ti, ta, te = (torch.tensor(x, dtype=O.DT) for x in (images, aux, eps))
    out, _ = O.loss_and_grads(p, ti, ta, te, beta=0.001, C_ma=torch.zeros((), dtype=O.DT),
                              lagrange_mult=torch.ones((), dtype=O.DT), alpha=0.0, kappa=math.sqrt(0.020),
                              clipping_qs=True, GECO=True, jitter=1e-6, N_train=N_TRAIN, L=L,
                              formulation="efficient")
    return float(out[0])


def cpu_worker_mnist(cfg3, budget_s, steps_only=0):
    """CHILD: the oracle timed on this host's cores (kind = "port": a restatement, TF1 cannot run here).  cfg2: the
    reference's own op sequence incl. the (b,m,m) tensor, explicit inverses and b x b diagonals ("literal"), then the
    O(L b m^2) efficient formulation for reference; cfg3: efficient only -- the literal form needs (b,m,m) tensors of
    0.5 GB per channel and operation."""
    from oracle import svgpvae_oracle as O
    B, M_IND, MDIM = (1024, 256, 32) if cfg3 else (256, 32, 8)
    params, images, aux, eps = synthetic_problem(0, B, M_IND, MDIM)
    p = {k: torch.tensor(np.asarray(v), dtype=O.DT) for k, v in params.items()}
    ti, ta, te = (torch.tensor(x, dtype=O.DT) for x in (images, aux, eps))
    ms = {k: torch.zeros_like(v) for k, v in p.items()}
    vs = {k: torch.zeros_like(v) for k, v in p.items()}
    kw = dict(beta=0.001, C_ma=torch.zeros((), dtype=O.DT), lagrange_mult=torch.ones((), dtype=O.DT), alpha=0.0,
              kappa=math.sqrt(0.020), clipping_qs=True, GECO=True, jitter=1e-6, N_train=N_TRAIN, L=L)
    ctr = [1]

    def one(form):
        _, g = O.loss_and_grads(p, ti, ta, te, formulation=form, **kw)
        O.adam_tf1_step(p, g, ms, vs, ctr[0], 1e-3)
        ctr[0] += 1

    def timed(form, budget, nmin):
        one(form)
        n, t0 = 0, time.perf_counter()
        while True:
            one(form)
            n += 1
            el = time.perf_counter() - t0
            if (el > budget and n >= nmin) or n >= 400 or (steps_only and n >= steps_only):
                return n, el

    main = "efficient" if cfg3 else "literal"
    n, el = timed(main, budget_s, 3)
    out = dict(value=n / el, n=n, seconds=el, formulation=main, threads=torch.get_num_threads(), rows=B)
    if not cfg3 and not steps_only:
        ne, ele = timed("efficient", 4.0, 3)
        out["efficient_formulation_steps_per_s"] = ne / ele
    print(json.dumps(out), flush=True)


def cpu_baseline_mnist(cfg3, budget_s=15.0):
    kind = "cfg3" if cfg3 else "cfg2"
    cands = sorted({host_threads(c) for c in ((16, 32, 64, 128) if cfg3 else (8, 16, 32, 64, 128))})
    probe = {nt: cpu_worker_call(kind, nt, 0.0, ("--probe-steps", "2"))["value"] for nt in cands} if len(cands) > 1 \
        else {cands[0]: 0.0}
    nt = max(probe, key=probe.get)
    r = cpu_worker_call(kind, nt, budget_s)
    out = dict(value=r["value"], unit="steps/s", cores=nt, host_cores=os.cpu_count(), threads=nt, kind="port",
               formulation=r["formulation"],
               sample=f"{r['n']} {r['formulation']}-formulation float64 steps (torch-CPU autograd + TF1 Adam) of the same "
                      f"{r['rows']}-row batch, {r['seconds']:.1f} s, {nt} OpenMP threads (fastest of {cands}) on a "
                      f"{os.cpu_count()}-core host")
    if "efficient_formulation_steps_per_s" in r:
        out["efficient_formulation_steps_per_s"] = r["efficient_formulation_steps_per_s"]
    return out


def mnist_survey_flops(b, L_, m, D):
    """one figure per step -- nets 0.49 MFLOP/img forward (x3 fwd + bwd), GP block
    F_gp = (5L+2) b m^2 + 5.33 L m^3 (x3) + kernel build (b m + m^2)(2D + 12) (x3).  0.45 GFLOP at config 2, 22 at config 3."""
    return 3 * 0.49e6 * b + 3 * ((5 * L_ + 2) * b * m * m + 5.33 * L_ * m ** 3) + 3 * (b * m + m * m) * (2 * D + 12)


def run_mnist(args):
    """One process group and one library communicator per process; then one measurement per scaling mode.  With several ranks
    and no explicit --scaling BOTH lines are printed: first the strong one (the headline configuration's global batch split over
    the ranks -- the same 256 / 1024 rows at every N, so its ELBO is one number for N = 1, 2, 4, 8: SURVEY 8e), then the weak one
    (rows per GPU fixed), which stays the LAST stdout line and carries the strong result as `strong_scaling`."""
    rank, local_rank, world = dist_env()
    import torch.distributed as dist
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    ctx = {"comm": None, "why": None}
    if (multi or args.force_comm) and args.exchange == "rccl":
        # the all-reduces are issued by the library on the compute stream (svgp_mnist_train_step_dp)
        ctx["comm"], ctx["why"] = library_comm(multi, local_rank, dev)
        if ctx["comm"] is None:      # both legs are RCCL; this only changes who enqueues the collective
            print(f"[bench] rank {rank}: in-library RCCL communicator unavailable ({ctx['why']}); "
                  f"using torch.distributed all_reduce between per-phase graphs", file=sys.stderr, flush=True)
    modes = [args.scaling] if args.scaling else (["strong", "weak"] if multi else ["weak"])
    strong = None
    for mode in modes:
        # Without an explicit --scaling the strong case is BEST EFFORT: it must never cost the run its headline (weak)
        # line.  A global batch the rank count does not divide is skipped with a note -- decided from the arguments alone, so
        # every rank skips together -- and an ELBO further than 1e-8 from the oracle is reported in the line, not asserted.
        best_effort = mode == "strong" and not args.scaling
        if best_effort:
            gb = args.global_batch or ((1024 if args.workload == "cfg3" else 256))
            if gb % world:
                if rank == 0:
                    print(f"[bench] strong-scaling case skipped: global batch {gb} is not divisible by {world} ranks",
                          file=sys.stderr, flush=True)
                continue
        line = mnist_case(args, mode, multi, dev, ctx, strict=not best_effort)
        if rank == 0:
            if mode == "strong":
                strong = {k: line.get(k) for k in ("value", "unit", "ms_per_step", "elbo", "elbo_rel_err_gpu_vs_oracle", "block_ms",
                                                   "elbo_within_1e-8_of_oracle")}
                strong["global_batch"] = line["config"]["global_batch"]
                strong["collectives_us_total"] = line.get("collectives_us_total")
            elif strong is not None:
                line["strong_scaling"] = strong
            emit(line)
    if multi:
        dist.destroy_process_group()


def mnist_case(args, scaling, multi, dev, ctx, strict=True):
    rank, local_rank, world = dist_env()
    import torch.distributed as dist
    cfg3 = args.workload == "cfg3"
    B, M_IND, MDIM = (1024, 256, 32) if cfg3 else (256, 32, 8)
    if scaling == "strong":
        # the SAME global batch at every rank count (rank 0's synthetic rows), split into equal contiguous row blocks
        gb = args.global_batch or B
        if gb % world:
            raise SystemExit(f"--global-batch {gb} is not divisible by {world} ranks")
        B = gb // world
        params, images, aux, eps = synthetic_problem(0, gb, M_IND, MDIM)
        images, aux, eps = (x[rank * B:(rank + 1) * B] for x in (images, aux, eps))
        global_rows = lambda: synthetic_problem(0, gb, M_IND, MDIM)[1:]
    else:
        params, images, aux, eps = synthetic_problem(rank, B, M_IND, MDIM)

        def global_rows():
            parts = [synthetic_problem(r, B, M_IND, MDIM) for r in range(world)]
            return tuple(np.concatenate([p[i] for p in parts], 0) for i in (1, 2, 3))

    from svgp_vae_amd.engine import MnistStepEngine
    eng = MnistStepEngine(M_IND, L, MDIM, N_OBJ, N_train=N_TRAIN, jitter=1e-6, clip_qs=True, geco=True,
                          kappa_squared=0.020, alpha=0.99, beta=0.001, lr=1e-3, b_max=B, device=str(dev),
                          rank=rank, world_size=world,
                          # --force-comm on one GPU: the kernel configuration of a multi-rank step (m > 64: the multi-rank
                          # workspace; m <= 64: row partials kept and exchanged, as the engine's multi-rank default) + a 1-rank
                          # communicator, i.e. everything of the N > 1 step but the wire
                          single_stat_block=((M_IND > 64 or os.environ.get("SVGP_DP_STAT_PARTIALS") == "0")
                                             if args.force_comm else None),
                          split_grad_exchange=args.split_grad)
    eng.load_params(params)
    d_img, d_aux, d_eps = (torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=dev).contiguous()
                           for x in (images, aux, eps))

    use_graph = not multi and not args.no_graph and not args.force_comm
    launch, comm_ranks = None, None
    if ctx["comm"] is not None:
        eng.attach_comm(ctx["comm"])
        comm_ranks = ctx["comm"].world_size
        launch = "one in-order stream: phases + in-library RCCL all-reduce x3"

    # ---- parity gate: ONE explicit-eps step (no optimiser update) through the same exchange path the timed region
    # uses; its ELBO is checked against the oracle's efficient formulation on the GLOBAL batch (all ranks' rows are
    # regenerated on rank 0 from their seeds).  north_star: "1/2/4/8-GPU ELBO equal at the same global batch".
    eng.bind(d_img, d_aux, d_eps)
    eng.run(adam=False)
    eng.synchronize()
    gpu_elbo = eng.scalars()["elbo"]
    eng.reset_state()
    elbo_rel = None
    if rank == 0 and not args.no_parity_gate:
        want = oracle_elbo(params, *global_rows())
        elbo_rel = abs(gpu_elbo - want) / abs(want)
        assert elbo_rel < 1e-3, f"ELBO parity failed at {world} rank(s): GPU {gpu_elbo} oracle {want}"
        # the strong line's global batch is the same rows at every rank count: agreeing with the oracle to 1e-8 at N = 1, 2, 4, 8
        # IS the cross-N equality of SURVEY 8e (the `elbo` field of the lines can also be compared directly)
        if scaling == "strong" and elbo_rel >= 1e-8:
            msg = f"strong-scaling ELBO at {world} rank(s): GPU {gpu_elbo} oracle {want} (rel {elbo_rel:.2e} >= 1e-8)"
            assert not strict, msg            # a hard failure only with an explicit --scaling strong
            print("[bench] " + msg, file=sys.stderr, flush=True)
    if multi:
        dist.barrier()

    # ---- timed region: eps drawn on device every step (tf.random.normal, SVGPVAE_model.py:901)
    eng.bind(d_img, d_aux, None)
    if use_graph:
        # one GPU: the step as ONE hipGraph replay or as eager launches through the C step entry point -- whichever is faster on
        # this box in an untimed probe (config 2: 14 launches per step keep the host far ahead and eager wins by ~2.5 %;
        # config 3: the replayed graph overlaps the two branches of its step slightly better and wins by ~3 %)
        eng.capture("step", adam=True)
        cand = {"hipGraph replay": (lambda: eng.replay("step")), "eager phases": (lambda: eng.run(adam=True))}
        probe = {}
        for name, fn in cand.items():
            probe[name] = min(timed_blocks(fn, eng.synchronize, max(20, args.steps // 4), 5, 3, False, dev))
        launch = min(probe, key=probe.get)
        step = cand[launch]
        launch += " (faster of hipGraph replay / eager phases in an untimed probe)"
    elif launch is not None:
        step = lambda: eng.run(adam=True)
    elif not args.no_graph and multi:
        # one hipGraph per phase, torch.distributed (RCCL) all-reduces (not captured) in between
        eng.capture_phases("step", adam=True)
        step = lambda: eng.run_phase_graphs("step")
        launch = "per-phase hipGraphs + torch.distributed RCCL all-reduce x3"
        comm_ranks = dist.get_world_size()
    else:
        step = lambda: eng.run(adam=True)
        launch = "eager phases" + (" + torch.distributed RCCL all-reduce x3" if multi else "")
    blocks = timed_blocks(step, eng.synchronize, args.steps, args.warmup, args.repeats, multi, dev)
    el = float(np.median(blocks))
    sc = eng.scalars()
    assert math.isfinite(sc["elbo"]) and sc["adam_t"] >= args.steps * args.repeats, sc
    # per-exchange-point microseconds (HIP events around pack + grouped collective + unpack, svgp_comm_timing): a few extra
    # steps after the timed region, the maximum over ranks of the per-point medians
    def collectives():
        eng.comm.timing(True)
        samples = []
        for _ in range(10):
            eng.run(adam=True)
            eng.synchronize()
            samples.append(eng.comm.timing_read())
        eng.comm.timing(False)
        med = torch.tensor(np.median(np.array(samples), 0), dtype=torch.float64, device=dev)
        if multi:
            dist.all_reduce(med, op=dist.ReduceOp.MAX)
        return [round(float(x), 1) for x in med.cpu()]

    coll_us, other = None, None
    if eng.comm is not None:
        coll_us = collectives()
        # With several ranks the other setting is timed only on request (--split-probe): it is code no multi-GPU box has run yet,
        # and it would sit between the timed region and the result line of the scaling run.  One rank (--force-comm /
        # --force-dist): always, unless --no-split-probe.
        if not eng.channel_sharded() and not args.no_split_probe and (world == 1 or args.split_probe):
            # the OTHER setting of cfg.split_grad_exchange (gradient all-reduce in two parts, the first beside the encoder's reverse
            # pass): a short timed block + its exchange points, so that the first multi-GPU run shows which one to keep
            flag = bool(eng.base["split_grad_exchange"])
            eng.set_split_grad_exchange(not flag)
            eng.bind(d_img, d_aux, None)
            ob = timed_blocks(lambda: eng.run(adam=True), eng.synchronize, max(20, args.steps // 3), 5, 3, multi, dev)
            other = {"split_grad_exchange": not flag, "ms_per_step": float(np.median(ob)) / max(20, args.steps // 3) * 1e3,
                     "collectives_us": collectives()}
            eng.set_split_grad_exchange(flag)
            eng.bind(d_img, d_aux, None)
    # per-stage HIP-event timings of this rank's launches (rank-local kernels, no collective inside): every rank runs
    # them so that nobody waits on rank 0, rank 0 reports
    stage_rows = time_stages(eng, B, M_IND, reps=(1 if os.environ.get("SVGP_BENCH_NO_STAGES") else 20 if cfg3 else 50))
    if multi:
        dist.barrier()

    if rank == 0:
        step_flops = sum(r["flops"] for r in stage_rows)
        name = ("BASELINE configs[1]: rotated-MNIST SVGPVAE_Hensman, m=32 inducing, L=16, GPLVM dim 8, batch 256 per "
                "GPU, N_train=4050, GECO + clip_qs, float64") if not cfg3 else \
               ("BASELINE configs[2]: rotated-MNIST SVGPVAE_Hensman, m=256 inducing, L=16, GPLVM dim 32 (SURVEY F9), "
                "batch 1024 per GPU, N_train=4050, GECO + clip_qs, float64, large-m GEMM path")
        line = {
            "metric": "SVGPVAE train steps/sec, rotated MNIST (N=4050, M=32, L=16)" if not cfg3 else
                      "SVGPVAE train steps/sec, rotated MNIST (N=4050, M=256, L=16, batch 1024)",
            "value": world * args.steps / el if scaling == "weak" else args.steps / el,
            "unit": f"steps/s ({B}-row batches, whole job)" if scaling == "weak" else
                    f"steps/s ({B * world}-row global batches)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "repeats": args.repeats, "block_ms": [round(x * 1e3, 3) for x in blocks], "timing": "median block",
            "config": {"workload": name, "global_batch": B * world, "rows_per_gpu": B, "launch": launch,
                       "parallelism": f"dp{world}", "rccl_ranks": comm_ranks,
                       "single_stat_block": int(eng.base["single_stat_block"]),
                       "scaling_note": ("weak: rows per GPU fixed, the global batch (and so c = N_train / b_global of "
                                        "SVGPVAE_model.py:328) grows with N" if scaling == "weak" else
                                        "strong: global batch fixed, rows per GPU = global / N")},
            "elbo_rel_err_gpu_vs_oracle": elbo_rel,
        }
        if scaling == "strong" and elbo_rel is not None:
            line["elbo_within_1e-8_of_oracle"] = bool(elbo_rel < 1e-8)
        top = stage_rows[0]
        # stage -> the kernel name rocprofv3 reports for it, where the two differ
        kname = {"gp_stats_factor_bwd": "k_gp_factor_bwd", "decoder_bwd_data": "k_decoder_bwd_data_aji" if M_IND <= 32 else
                 "k_decoder_bwd_data"}.get(top["stage"], "k_" + top["stage"])
        traffic, src = committed_traffic(kname) if not cfg3 else (None, None)
        roof = roofline_of(top["flops"], top["bytes"], top["us"], F64_PEAK_TFLOPS, kernel=top["stage"],
                           implementation_partials_bytes=top["partials_bytes"],
                           note="config 2 is latency-bound (DESIGN.md section 5); " * (not cfg3) +
                                "stage = one C entry point, HIP events on the launch stream")
        roof["traffic"], roof["traffic_source"] = traffic, src
        line["roofline"] = roof
        sf = mnist_survey_flops(B, L, M_IND, MDIM + 1)
        line["step_roofline"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": F64_PEAK_TFLOPS,
                                 "algorithmic_flops": sf,
                                 "achieved": sf / (el / args.steps) / 1e12,
                                 "frac": sf / (el / args.steps) / 1e12 / F64_PEAK_TFLOPS,
                                 "flops_model": "SURVEY 8d / Appendix G (the one figure this file uses for the step)",
                                 "stage_sum_flops": step_flops}
        line["step_roofline"]["traffic"], line["step_roofline"]["traffic_source"] = committed_step_traffic("cfg3" if cfg3 else "cfg2")
        line["stages_us"] = {r["stage"]: round(r["us"], 2) for r in stage_rows}
        line["step_flops"] = step_flops
        if coll_us is not None:
            def named(us):
                names = {5: ["rs[S|v]", "ag[Si|t|u]", "rs[A2|ud|td]", "ag[Ssym|vbar|KL]", "ar[grad|sums]"],
                         4: ["ar[S|v]", "ar[A2|ud|td]", "ar[grad tail|sums] (side branch)", "ar[grad head]"],
                         3: ["ar[S|v]", "ar[A2|ud|td]", "ar[grad|sums]"]}[len(us)]
                return dict(zip(names, us))
            line["collectives_us"] = named(coll_us)
            line["collectives_us_total"] = round(sum(coll_us), 1)
            line["config"]["split_grad_exchange"] = bool(eng.base["split_grad_exchange"])
            if other is not None:
                other["collectives_us"] = named(other["collectives_us"])
                line["other_exchange_setting"] = other
            line["collectives_note"] = ("HIP events on the compute stream around every exchange point (pack + ONE grouped RCCL "
                                        "launch + unpack), median of 10 steps, maximum over ranks; compute = ms_per_step - total")
        if world == 1 and not args.no_cpu_baseline and scaling == "weak":
            line["cpu_baseline"] = cpu_baseline_mnist(cfg3)
        line["elbo"] = gpu_elbo
        return line
    return None


# =====================================================================================================================
# sprites800: BASELINE configs[3] shape, one GPU's share (500 frames, L=64, m=800)
# =====================================================================================================================
def sprites_problem(rank, b, L_, La, Lc, n_act, m, seed=0):
    rs = np.random.RandomState(seed)
    ip = rs.normal(0, 1.5, (m, La + Lc))
    table = rs.normal(0, 1.5, (n_act, La))
    g = torch.Generator().manual_seed(rank)
    img = torch.rand(b, 64, 64, 3, dtype=torch.float64, generator=g)
    ids = torch.tensor(np.random.RandomState(rank).randint(0, n_act, b), dtype=torch.float64)
    eps = torch.randn(b, L_, dtype=torch.float64, generator=g)
    return ip, table, img, ids, eps


def sprites_flops(b, L_, m, D=24):
    """nets 35 MMAC/frame fwd (x3 fwd+bwd), GP block F_gp = (5L+2) b m^2 + 5.33 L m^3 (x3) + kernel build."""
    nets = 3 * 2 * 35.0e6 * b
    gp = 3 * ((5 * L_ + 2) * b * m * m + 5.33 * L_ * m ** 3) + 3 * (b * m + m * m) * (2 * D + 12)
    return nets, gp


# --kernel se (the reference's --K_SE, SVGPVAE_model.py:530-544): l_action, sigma_action, l_character, sigma_character -- length
# scales of the order of the synthetic vectors' distances, so that K_mm is full rank with entries across (0, 1)
SPRITES_SE = (6.0, 1.0, 6.0, 0.8)


def run_sprites(args):
    rank, local_rank, world = dist_env()
    import torch.distributed as dist
    from svgp_vae_amd import sprites as S
    from svgp_vae_amd.engine import RcclComm
    multi = world > 1 or args.force_dist          # --force-dist: process group + communicator + exchange points with ONE rank
    comm, comm_why = None, None
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        # the library's communicator when EVERY rank can create one (collective MIN vote), torch.distributed otherwise
        from svgp_vae_amd.dp import TorchDistComm
        comm, comm_why = library_comm(True, local_rank, torch.device(f"cuda:{local_rank}"))
        if comm is None:
            print(f"[bench] rank {rank}: in-library RCCL communicator unavailable ({comm_why}); exchanging through "
                  f"torch.distributed", file=sys.stderr, flush=True)
            comm = TorchDistComm()
    elif args.force_comm:
        comm = RcclComm(0, 1, RcclComm.unique_id())
    # several ranks and no explicit --scaling: the strong line first (the configuration's own 500 frames cut in whole 50-frame
    # character groups -- one ELBO at every rank count, checked against the oracle), then the weak line, which stays last
    modes = [args.scaling] if args.scaling else (["strong", "weak"] if multi and world <= 10 else ["weak"])
    for scaling in modes:
        sprites_case(args, scaling, multi, comm, comm_why)
    if multi:
        dist.destroy_process_group()


def sprites_case(args, scaling, multi, comm, comm_why):
    rank, local_rank, world = dist_env()
    import torch.distributed as dist
    from svgp_vae_amd import sprites as S
    frames, L_, La, Lc, n_act, m = 50, 64, 8, 16, 72, args.m or 800
    if scaling == "strong":
        # the SAME 500 frames (rank 0's synthetic batch) at every rank count, cut in whole 50-frame character groups
        from svgp_vae_amd.dp import shard_batch
        gb = args.global_batch or 500
        lo, hi = shard_batch(0, gb, world, rank, frames)
        if hi <= lo:
            raise SystemExit(f"--scaling strong: {gb} frames leave rank {rank} of {world} without a character group")
        ip, table, img, ids, eps = sprites_problem(0, gb, L_, La, Lc, n_act, m)
        img, ids, eps = img[lo:hi], ids[lo:hi], eps[lo:hi]
        b, b_global, b_cap = hi - lo, gb, -(-(gb // frames) // world) * frames
    else:
        b = b_global = b_cap = 500
        b_global = b * world
        ip, table, img, ids, eps = sprites_problem(rank, b, L_, La, Lc, n_act, m)
    k_se = args.kernel == "se"
    svgp = S.spritesSVGP(False, False, ip, 'main', 0.01, 50000, La, table, Lc, L_, K_obj_normalize=True, K_SE=k_se)
    if k_se:
        svgp.se = torch.tensor(SPRITES_SE, dtype=torch.float64)
    f32 = args.precision == "f32"
    # f32: the networks in float32 (the reference's dtype, VAE_utils.py:277); the GP block stays float64 unless --gemm-f32
    # asks otherwise (float32 products lose parity / stability at m = 800: tests/test_gpu_f32.py)
    eng = S.SpritesStepEngine(S.spritesVAE(L_), S.sprites_representation_network(Lc), svgp, b_max=b_cap, seg_len=frames,
                              geco=True, kappa_squared=0.0075, clip_grad=1e6, device=f"cuda:{local_rank}", rank=rank,
                              world_size=world, comm=comm, net_dtype=torch.float32 if f32 else torch.float64,
                              gemm_f32=args.gemm_f32,
                              # --force-comm on one GPU: the channel-sharded sequence of a multi-rank step (five grouped exchange
                              # points, tile-packed blocks, window = all channels) through a 1-rank communicator
                              channel_shard=True if (args.force_comm and not multi) else None)
    dev = eng.dev
    d_img, d_ids, d_eps = img.to(dev, eng.ndt), ids.to(dev), eps.to(dev)
    # parity gate (N = 1): explicit-eps step, ELBO against the oracle's efficient formulation -- inside cpu_baseline
    eng.step(d_img, d_ids, d_eps, adam=False, b_global=b_global)
    gpu_elbo = eng.scalars()["elbo"]
    eng.set_scalars(c_ma=0.0, lagrange=1.0, alpha=0.0)
    step = lambda: eng.step(d_img, d_ids, None, adam=True, b_global=b_global)
    blocks = timed_blocks(step, eng.stream.synchronize, args.steps, args.warmup, args.repeats, multi, dev)
    el = float(np.median(blocks))
    # ---- stage times: HIP events recorded on the engine's stream at the stage boundaries of one more step
    eng.trace = []
    step()
    eng.stream.synchronize()
    marks, eng.trace = eng.trace, None
    stages = {}
    for (n0, e0), (_, e1) in zip(marks[:-1], marks[1:]):
        stages[n0] = stages.get(n0, 0.0) + e0.elapsed_time(e1) * 1e3
    # ... and the same step issued on ONE stream (side branches switched off: the same launches in program order, bit-identical
    # results -- tests/test_gpu_sprites.py), where a stage group's time is the time of its own launches; inside the three-stream
    # step the groups share the chip with the side branches' GEMMs and their wall time says how the step was scheduled
    stages_alone = None
    if getattr(eng, "side", None) is not None and not os.environ.get("SVGP_BENCH_NO_STAGES"):
        keep = (eng.side, eng.side2)
        eng.side = eng.side2 = None
        step(); eng.stream.synchronize()
        eng.trace = []
        step()
        eng.stream.synchronize()
        marks1, eng.trace = eng.trace, None
        eng.side, eng.side2 = keep
        stages_alone = {}
        for (n0, e0), (_, e1) in zip(marks1[:-1], marks1[1:]):
            stages_alone[n0] = stages_alone.get(n0, 0.0) + e0.elapsed_time(e1) * 1e3
    sc = eng.scalars()
    assert math.isfinite(sc["elbo"]), sc
    # per-exchange-point microseconds (events around pack + grouped RCCL launch + unpack), maximum over ranks
    coll_us = None
    if comm is not None:
        samples = []
        for _ in range(3):
            eng.exchange_trace = []
            step()
            eng.stream.synchronize()
            samples.append([e0.elapsed_time(e1) * 1e3 for e0, e1 in eng.exchange_trace])
        eng.exchange_trace = None
        med = torch.tensor(np.median(np.array(samples), 0), dtype=torch.float64, device=dev)
        if multi:
            dist.all_reduce(med, op=dist.ReduceOp.MAX)
        coll_us = [round(float(x), 1) for x in med.cpu()]
    if rank == 0:
        nets, gp = sprites_flops(b, L_, m)
        ms = el / args.steps * 1e3
        line = {
            "metric": f"SVGPVAE train steps/sec, SPRITES 64x64 (m={m}, L=64, 500 frames per GPU)" if scaling == "weak" else
                      f"SVGPVAE train steps/sec, SPRITES 64x64 (m={m}, L=64, {b_global} frames per step over all GPUs)",
            "value": world * args.steps / el if scaling == "weak" else args.steps / el,
            "unit": "steps/s (500-frame batches, whole job)" if scaling == "weak" else f"steps/s ({b_global}-frame global batches)",
            "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": eng.dtype_name, "data": "synthetic",
            "repeats": args.repeats, "block_ms": [round(x * 1e3, 3) for x in blocks], "timing": "median block",
            "config": {"workload": f"BASELINE configs[3] shape: SPRITES SVGPVAE_Hensman + GPLVM, {b} frames per GPU "
                                   f"(10 characters x 50), L=64, L_action=8, L_character=16, m={m}, jitter 0.01, "
                                   + ("SE x SE kernel (--K_SE; full-rank K_mm)" if k_se else "cosine-normalised linear x linear kernel")
                                   + ", GECO, gradient clip 1e6",
                       "global_batch": b_global, "rows_per_gpu": b, "parallelism": f"dp{world}",
                       "rccl_ranks": None if (comm is None or comm_why is not None) else comm.world_size,
                       "comm_fallback": comm_why,
                       "exchange": None if comm is None else (
                           "channel-sharded, one grouped RCCL launch per point, symmetric blocks tile-packed: reduce-scatter "
                           "S,v | all-gather Sigma^-1,M2,t,u | reduce-scatter A2,ud,td | all-gather Ssym,vbar,KL | all-reduce "
                           "gradients" if eng.chan_shard else "all-reduce x3"),
                       "launch": "eager stream" + ("" if comm is None else " + in-library RCCL collectives")},
        }
        peak = F64_PEAK_TFLOPS          # the dominant stage groups are the float64 GP factor stages in both precisions
        top = max(stages, key=stages.get)
        gflops = {"gp_fwd": gp / 3, "gp_bwd": 2 * gp / 3, "nets_fwd": nets / 3, "nets_bwd": 2 * nets / 3}
        grp = "gp_" + ("fwd" if "fwd" in top else "bwd") if top.startswith("gp") else \
              "nets_" + ("fwd" if "fwd" in top else "bwd")
        grp_us = sum(v for k, v in stages.items() if (k.startswith("gp") == grp.startswith("gp")) and
                     (("fwd" in k) == ("fwd" in grp)))
        # algorithmic bytes of a stage group: every (L,m,m) float64 intermediate of the group written once and read once (forward
        # S, Sigma^-1, G, A_hat, M2, (A_hat + jI)^-1; reverse: twice as many), the (L,b,m) row products likewise, the networks'
        # activations once each way -- 70 flop per byte for the GP groups at m = 800, far on the MFMA side of the ridge
        blk, rowp = 8.0 * L_ * m * m, 8.0 * L_ * b * m
        act = (4.0 if f32 else 8.0) * b * 64 * 64 * (3 + 7 * 16 + 5 * 16 / 4 + 3 * 16 / 16)     # rough: 64^2 layers dominate
        gbytes = {"gp_fwd": 2 * 6 * blk + 2 * 2 * rowp, "gp_bwd": 2 * 12 * blk + 2 * 2 * rowp, "nets_fwd": 2 * act,
                  "nets_bwd": 4 * act}
        line["roofline"] = roofline_of(gflops[grp], gbytes[grp], grp_us, peak, kernel=grp,
                                       note="stage group = the launches between two HIP events on the engine stream; "
                                            "algorithmic flops of SURVEY 8d split 1/3 forward, 2/3 reverse; bytes: every "
                                            "intermediate of the group written once and read once")
        traffic, src = committed_traffic("sprites800_" + grp)
        line["roofline"]["traffic"], line["roofline"]["traffic_source"] = traffic, src
        line["step_roofline"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": peak, "algorithmic_flops": nets + gp,
                                 "achieved": (nets + gp) / (ms * 1e-3) / 1e12,
                                 "frac": (nets + gp) / (ms * 1e-3) / 1e12 / peak,
                                 "frac_of_f32_peak": (nets + gp) / (ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS}
        line["step_roofline"]["traffic"], line["step_roofline"]["traffic_source"] = committed_step_traffic("sp800")
        line["stages_us"] = {k: round(v, 1) for k, v in sorted(stages.items(), key=lambda kv: -kv[1])}
        if stages_alone is not None:
            line["stages_one_stream_us"] = {k: round(v, 1) for k, v in sorted(stages_alone.items(), key=lambda kv: -kv[1])}
            line["nets_ms"] = {"in_step": round(sum(v for k, v in stages.items() if k.startswith("nets")) / 1e3, 3),
                               "one_stream": round(sum(v for k, v in stages_alone.items() if k.startswith("nets")) / 1e3, 3),
                               "note": "sum of the four network stage groups; in_step = wall time between stage events while the "
                                       "side branches' GEMMs share the chip, one_stream = the same launches with nothing beside them"}
        if coll_us is not None:
            names = (["rs[S|v]", "ag[Si|t|u]", "rs[A2|ud|td]", "ag[Ssym|vbar|KL]", "ar[grad|sums]"] if len(coll_us) == 5
                     else ["ar[S|v]", "ar[A2|ud|td]", "ar[grad|sums]"])
            line["collectives_us"] = dict(zip(names, coll_us))
            line["collectives_us_total"] = round(sum(coll_us), 1)
        # the oracle step is on rank 0's 500 frames: the batch of the one-rank run and of the strong case at any rank count
        if (world == 1 or (scaling == "strong" and b_global == 500)) and not args.no_cpu_baseline:
            cb = cpu_baseline_sprites(gpu_elbo, m, k_se)
            line["elbo_rel_err_gpu_vs_oracle"] = cb.pop("elbo_rel_err_gpu_vs_oracle")
            if world == 1 and scaling == "weak":
                line["cpu_baseline"] = cb
        emit(line)


def cpu_worker_sprites(m, k_se=False):
    """CHILD: ONE efficient-formulation float64 step of the oracle (torch-CPU autograd) on the same 500-frame batch: the
    literal form's (b,m,m) tensors are 2.6 GB per channel x 64 channels."""
    from oracle import sprites_oracle as SO
    DT = torch.float64
    b, frames, L_, La, Lc, n_act = 500, 50, 64, 8, 16, 72
    ip, table, img, ids, eps = sprites_problem(0, b, L_, La, Lc, n_act, m)
    params = {k: torch.as_tensor(np.asarray(v), dtype=DT) for k, v in SO.glorot_init(L_, Lc, 0).items()}
    gp = dict(inducing_index_points=torch.tensor(ip, dtype=DT), GPLVM_action=torch.tensor(table, dtype=DT),
              **{k: torch.tensor(v if k_se else 1.0, dtype=DT)
                 for k, v in zip(("l_action", "sigma_action", "l_character", "sigma_character"), SPRITES_SE)})
    seg, rep = SO.aux_data_sprites_utils(b, frames, frames)
    kw = dict(beta=0.001, C_ma=torch.tensor(0.0, dtype=DT), lagrange_mult=torch.tensor(1.0, dtype=DT), alpha=0.0,
              kappa=math.sqrt(0.0075), L=L_, L_action=La, jitter=0.01, N_train=50000.0, segment_ids=seg, repeats=rep,
              clipping_qs=False, GECO=True, K_obj_normalize=True, K_SE=k_se, clip_grad=1e6, titsias=False)
    t0 = time.perf_counter()
    want, _ = SO.loss_and_grads(params, gp, (img, ids.long()), eps, formulation="efficient", **kw)
    el = time.perf_counter() - t0
    print(json.dumps(dict(seconds=el, elbo=float(want[0]), threads=torch.get_num_threads(), rows=b)), flush=True)


def cpu_baseline_sprites(gpu_elbo, m, k_se=False):
    nt = host_threads(32)
    r = cpu_worker_call("sprites800", nt, 0.0, ("--m", str(m), "--kernel", "se" if k_se else "linear"))
    rel = abs(gpu_elbo - r["elbo"]) / abs(r["elbo"])
    assert rel < 1e-3, f"ELBO parity failed: GPU {gpu_elbo} oracle {r['elbo']}"
    return dict(value=1.0 / r["seconds"], unit="steps/s", cores=nt, host_cores=os.cpu_count(), threads=nt, kind="port",
                formulation="efficient", elbo_rel_err_gpu_vs_oracle=rel,
                sample=f"1 efficient-formulation float64 forward + autograd reverse of the same {r['rows']}-frame batch "
                       f"(no Adam update), {r['seconds']:.1f} s, {nt} OpenMP threads on a {os.cpu_count()}-core host")


# =====================================================================================================================
# cfg5: BASELINE configs[4], one GPU's shard: N = 131072 rows, m = 2048, L = 16, float32
# =====================================================================================================================
def run_cfg5(args):
    rank, local_rank, world = dist_env()
    import torch.distributed as dist
    from svgp_vae_amd import stream_stats as SS
    multi = world > 1 or args.force_dist          # --force-dist: process group + communicator + the S, v all-reduce with ONE rank
    comm, comm_why = None, None
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        from svgp_vae_amd.dp import TorchDistComm
        comm, comm_why = library_comm(True, local_rank, torch.device(f"cuda:{local_rank}"))
        if comm is None:
            print(f"[bench] rank {rank}: in-library RCCL communicator unavailable ({comm_why}); all-reducing through "
                  f"torch.distributed", file=sys.stderr, flush=True)
            comm = TorchDistComm()
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    n, m, L_, M = args.rows or 131072, args.m or 2048, 16, 8
    n_obj = 400
    kd = SS.kernel_desc(SS.PERIODIC_LINEAR, 2, M, n_table=n_obj, params=(1.0, 1.0))
    g, tab, x, z, means, vars_ = cfg5_inputs(rank, n, m, L_, M, n_obj)
    tab, x, z, means, vars_ = (t.to(dev) for t in (tab, x, z, means, vars_))
    D = 2 + M
    fi = SS.features(kd, z, inducing=True)
    K = torch.empty((n, m), dtype=torch.float32, device=dev)
    ws = SS.stats_workspace(n, m, L_, dev)
    S = torch.empty((L_, m, m), dtype=torch.float32, device=dev)
    v = torch.empty((L_, m), dtype=torch.float32, device=dev)
    fr_box = [SS.features(kd, x, inducing=False, table=tab)]

    def step():      # one pass of the N-sized statistics: features -> K_nm (materialised, 5a) -> S_l, v_l (5b) [-> all-reduce]
        fr_box[0] = SS.features(kd, x, inducing=False, table=tab)
        SS.knm(kd, fr_box[0], n, fi, m, out=K)
        SS.stats(K, means, vars_, ws=ws, S=S, v=v, comm=comm)

    blocks = timed_blocks(step, torch.cuda.synchronize, args.steps, args.warmup, args.repeats, multi, dev)
    el = float(np.median(blocks))

    def ev_time(fn, reps=5):         # SS.* launch on torch's current stream: torch events see them
        fn(); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for i in range(reps):
            fn(); ev[i + 1].record()
        torch.cuda.synchronize()
        return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2] * 1e3

    us_feat = ev_time(lambda: SS.features(kd, x, inducing=False, table=tab))
    us_knm = ev_time(lambda: SS.knm(kd, fr_box[0], n, fi, m, out=K))
    us_stats = ev_time(lambda: SS.stats(K, means, vars_, ws=ws, S=S, v=v))
    # parity probe at the full size (float64 GEMV chain): S_l w == K^T (p_l * (K w)), v_l == K^T (p_l * mean_l)
    w = torch.randn(m, generator=g).to(dev).double()
    Kw = K.double() @ w
    p = torch.where(vars_ == 0, torch.zeros_like(vars_), 1.0 / vars_).double()
    SS.stats(K, means, vars_, ws=ws, S=S, v=v)
    err = 0.0
    for l in range(L_):
        ref = K.t().double() @ (p[:, l] * Kw)
        err = max(err, float(((S[l].double() @ w) - ref).abs().max() / ref.abs().max()))
    assert err < 1e-4, err
    if rank == 0:
        alg_flops = float(L_) * n * m * m          # symmetric count (L N m^2)
        knm_bytes = 4.0 * (n * m + n * D + m * D)
        line = {
            "metric": "SVGP statistics pass rows/sec (N=2^20 frames, m=2048, L=16, fp32; K_nm build + S_l, v_l)",
            "value": world * n * args.steps / el, "unit": "rows/s (whole job)", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "repeats": args.repeats,
            "block_ms": [round(x * 1e3, 3) for x in blocks], "timing": "median block",
            "config": {"workload": f"BASELINE configs[4], one GPU's shard: {n} rows x m={m} inducing, L={L_}, "
                                   f"periodic x linear kernel (D={D}), float32: features, materialised K_nm (5a), "
                                   f"S_l / v_l statistics (5b)" + (", RCCL all-reduce of S, v" if comm is not None else ""),
                       "rows_per_gpu": n, "parallelism": f"dp{world}",
                       "rccl_ranks": None if (comm is None or comm_why is not None) else comm.world_size,
                       "comm_fallback": comm_why},
            "probe_rel_err_S": err,
        }
        t_s, src_s = committed_traffic("k_stats_mfma_f32", per_pass_of="k_stats_reduce_f32")
        line["roofline"] = roofline_of(alg_flops, 4.0 * (n * m + 2 * n * L_ + L_ * m * m), us_stats, F32_PEAK_TFLOPS,
                                       kernel="stream_stats_f32 (k_stats_mfma_f32 + v_l + reduction)",
                                       note="algorithmic flops L N m^2 (symmetric count, SURVEY 8d); executed 1.03x (diagonal tiles: upper 32 x 32 blocks only)")
        line["roofline"]["traffic"], line["roofline"]["traffic_source"] = t_s, src_s
        t_k, src_k = committed_traffic("k_knm_f32")
        line["roofline_knm"] = roofline_of((2 * D + 8) * n * m, knm_bytes, us_knm, F32_PEAK_TFLOPS,
                                           kernel="stream_knm_f32 (materialised K_nm, HBM-write-bound)")
        line["roofline_knm"]["traffic"], line["roofline_knm"]["traffic_source"] = t_k, src_k
        line["stages_us"] = {"features": round(us_feat, 1), "knm": round(us_knm, 1), "stats": round(us_stats, 1)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_cfg5(m, n)
        emit(line)
    if multi:
        dist.destroy_process_group()


def cfg5_inputs(rank, n, m, L_, M, n_obj=400):
    g = torch.Generator(device="cpu").manual_seed(rank)
    tab = torch.randn(n_obj, M, generator=g) * 1.5
    x = torch.cat([torch.randint(0, n_obj, (n, 1), generator=g).float(), torch.rand(n, 1, generator=g) * 6.2832,
                   torch.randn(n, M, generator=g)], 1).contiguous()
    z = torch.cat([torch.zeros(m, 1), torch.rand(m, 1, generator=g) * 6.2832,
                   torch.randn(m, M, generator=g) * 1.5], 1).contiguous()
    means = torch.randn(n, L_, generator=g)
    vars_ = torch.rand(n, L_, generator=g) * 9.999 + 1e-3
    return g, tab, x, z, means, vars_


def cpu_worker_cfg5(m, budget_s):
    """CHILD: float32 torch-CPU restatement of the same pass (kernel matrix of mnistSVGP.kernel_matrix, then per channel
    K_mn (K_nm / var_l), K_mn (mean_l / var_l), SVGPVAE_model.py:1004-1017) on a bounded row sample."""
    L_, M, rows, n = 16, 8, 4096, 32768
    _, tb, xs, zs, mu, var = cfg5_inputs(0, n, m, L_, M)

    def one(lo):
        xr = xs[lo:lo + rows]
        o = tb[xr[:, 0].long()]
        d = xr[:, 1:2] - zs[:, 1][None, :]
        K = torch.exp(-2.0 * torch.sin(0.5 * d) ** 2) * (o @ zs[:, 2:].t())
        p = 1.0 / var[lo:lo + rows]
        S = torch.stack([(K * p[:, l:l + 1]).t() @ K for l in range(L_)])
        vv = K.t() @ (p * mu[lo:lo + rows])
        return S, vv

    one(0)
    k, t0 = 0, time.perf_counter()
    while True:
        one((k * rows) % (n - rows + 1))
        k += 1
        el = time.perf_counter() - t0
        if el > budget_s and k >= 2:
            break
    print(json.dumps(dict(value=k * rows / el, chunks=k, rows=rows, seconds=el, threads=torch.get_num_threads())), flush=True)


def cpu_baseline_cfg5(m, n_rows, budget_s=12.0):
    nt = host_threads(64)
    r = cpu_worker_call("cfg5", nt, budget_s, ("--m", str(m)))
    return dict(value=r["value"], unit="rows/s", cores=nt, host_cores=os.cpu_count(), threads=nt, kind="port",
                sample=f"{r['chunks']} chunks of {r['rows']} rows (the GPU shard has {n_rows}) through float32 torch-CPU "
                       f"K_nm + S_l, v_l for m={m}, L=16, {r['seconds']:.1f} s, {nt} OpenMP threads on a "
                       f"{os.cpu_count()}-core host")


# =====================================================================================================================
def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n, argv, launcher_module=None):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child process, pass its stdout through line by
    line, and return its exit code.  The benchmark's JSON line(s) are remembered; if anything else (an RCCL banner, a warning
    of the launcher) followed the last one on stdout, it is printed again so that it is the parent's LAST stdout line.
    (The reference is single-process, MNIST_experiment.py:299,308 -- this launcher is the build's own)
    SVGP_BENCH_LAUNCHER names another launcher module (the CPU test substitutes a stub for torch.distributed.run)."""
    import subprocess
    mod = launcher_module or os.environ.get("SVGP_BENCH_LAUNCHER", "torch.distributed.run")
    # torch.distributed.run scans the script's flags as well and rejects `--m` (a prefix of its --max-restarts / --master-addr / ...)
    argv = ["--inducing" if a == "--m" else ("--inducing=" + a[4:] if a.startswith("--m=") else a) for a in argv]
    cmd = [sys.executable, "-m", mod, "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: the only mode the host driver supports
    env.setdefault("OMP_NUM_THREADS", "8")                  # torchrun would set 1 and warn; the CPU parity gate uses the oracle
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    last_json, last_line = None, None
    for raw in proc.stdout:
        txt = raw.rstrip("\n")
        if not txt.strip():
            continue
        print(txt, flush=True)
        last_line = txt
        if txt.lstrip().startswith("{") and '"metric"' in txt:
            last_json = txt
    rc = proc.wait()
    if rc == 0 and last_json is None:
        print("[bench] the launched ranks printed no result line", file=sys.stderr, flush=True)
        return 1
    if rc != 0:      #  a failed run must not leave a remembered result line of an EARLIER case as its last stdout line
        print(f"[bench] the launched ranks exited with code {rc}; no result line is re-emitted", file=sys.stderr, flush=True)
        return rc
    if last_json is not None and last_line != last_json:
        print(last_json, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--repeats", type=int, default=5, help="blocks of --steps steps; the median block is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--exchange", choices=["rccl", "torch"], default="rccl",
                    help="N>1: who enqueues the three all-reduces -- the library on the compute stream (rccl) or "
                         "torch.distributed between per-phase graphs (torch); both are RCCL over xGMI")
    ap.add_argument("--force-dist", action="store_true",
                    help="N=1 under torchrun: take the N>1 code path (process group, communicator bootstrap through "
                         "broadcast_object_list, barriers, MAX all-reduce of the time) with world size 1")
    ap.add_argument("--split-grad", action="store_true",
                    help="cfg2 / cfg3 with several ranks: cfg.split_grad_exchange -- the gradient all-reduce in two parts, the first "
                         "beside the encoder's reverse pass (the line reports the other setting too: other_exchange_setting)")
    ap.add_argument("--no-split-probe", action="store_true", help="do not time the other setting of --split-grad")
    ap.add_argument("--split-probe", action="store_true",
                    help="several ranks: also time the OTHER setting of cfg.split_grad_exchange after the timed region and report it as "
                         "other_exchange_setting (with one rank and a communicator this is the default)")
    ap.add_argument("--force-comm", action="store_true",
                    help="N=1: run the data-parallel entry point with a 1-rank communicator (plumbing check)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="cfg2 / cfg3: default = weak on one rank; with several ranks BOTH lines are printed (strong first, the "
                         "weak line last with the strong result inside as `strong_scaling`)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="--scaling strong: rows of the fixed global batch (default: the configuration's own batch, 256 / 1024)")
    ap.add_argument("--workload", choices=["cfg2", "cfg3", "sprites800", "cfg5"], default="cfg2",
                    help="cfg2 = BASELINE configs[1] (the metric's configuration, default); cfg3 = configs[2] "
                         "(m=256, b=1024, GPLVM dim 32); sprites800 = configs[3] shape on one GPU's share; "
                         "cfg5 = configs[4] shard (N=131072, m=2048, float32 statistics pass)")
    ap.add_argument("--m", "--inducing", dest="m", type=int, default=None,
                    help="sprites800 / cfg5: inducing points (default 800 / 2048); --inducing is the spelling the self-launcher "
                         "passes on (torch.distributed.run's own parser rejects --m as an ambiguous abbreviation)")
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64",
                    help="sprites800: f64 = everything float64; f32 = the three networks in float32 (the reference's "
                         "dtype), GP block float64")
    ap.add_argument("--gemm-f32", type=int, choices=[0, 1, 2], default=0,
                    help="sprites800: cfg.gemm_f32 of the large-m GP block (1 = every product on the float32 MFMA, "
                         "2 = the statistics products only); both lose stability at m = 800 (DESIGN.md)")
    ap.add_argument("--kernel", choices=["linear", "se"], default="linear",
                    help="sprites800: linear = cosine-normalised linear x linear kernels (the reference's default); se = the "
                         "reference's --K_SE (SE x SE, SVGPVAE_model.py:530-544)")
    ap.add_argument("--rows", type=int, default=None, help="cfg5: rows per GPU (default 131072)")
    ap.add_argument("--cpu-worker", choices=["cfg2", "cfg3", "sprites800", "cfg5"], default=None,
                    help="internal: run one CPU-baseline leg in this (child) process and print its JSON")
    ap.add_argument("--budget", type=float, default=15.0)
    ap.add_argument("--probe-steps", type=int, default=0)
    args = ap.parse_args()
    if args.cpu_worker is not None:
        if args.cpu_worker in ("cfg2", "cfg3"):
            cpu_worker_mnist(args.cpu_worker == "cfg3", args.budget, args.probe_steps)
        elif args.cpu_worker == "sprites800":
            cpu_worker_sprites(args.m or 800, args.kernel == "se")
        else:
            cpu_worker_cfg5(args.m or 2048, args.budget)
        return
    dflt = {"cfg2": (300, 30), "cfg3": (30, 5), "sprites800": (3, 1), "cfg5": (3, 1)}[args.workload]
    args.steps = dflt[0] if args.steps is None else args.steps
    args.warmup = dflt[1] if args.warmup is None else args.warmup
    if args.workload in ("sprites800", "cfg5") and args.repeats > 3:
        args.repeats = 3

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.force_dist):
        # plain `python bench.py --gpus N`: this process becomes the launcher.  Nothing here has touched the GPU yet
        # (argparse + imports only), and the ranks are CHILD processes -- never an exec of this one.
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    # dmabuf IPC is the only mode this pool's host driver supports (RCCL / cross-process tensor sharing fail without it); it is read
    # when the HIP runtime initialises, so it is set BEFORE the first GPU call of this process
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, local_rank, world = dist_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    torch.cuda.set_device(local_rank)
    {"cfg2": run_mnist, "cfg3": run_mnist, "sprites800": run_sprites, "cfg5": run_cfg5}[args.workload](args)


if __name__ == "__main__":
    main()
