"""Rotated-MNIST experiment driver with the reference's CLI (MNIST_experiment.py:1115-1174 flags and
defaults) for `--elbo SVGPVAE_Hensman | SVGPVAE_Titsias | SVIGP_Hensman`, running every step on the HIP library.

    python -m svgp_vae_amd.MNIST_experiment --elbo SVGPVAE_Hensman --ip_joint --GP_joint --ov_joint \
        --clip_qs --GECO --PCA --mnist_data_path "MNIST data/"

Mirrors run_experiment_rotated_mnist_SVGPVAE (MNIST_experiment.py:30-541): model construction with the
inverted `fixed_*` flags (:96-98), the un-shuffled epoch loop with ragged last batch (:319-355), GECO state
carry / first-step alpha=0 (kept on device), and every 10 epochs: eval-set reconstruction MSE, conditional
generation MSE on the test set (:457-486) and `pics/test_metrics.txt` lines `epoch,recon MSE,cgen MSE`
(:509-510).  Plotting / pandas logging of the reference are not reproduced.
`--elbo SVIGP_Hensman` mirrors run_experiment_rotated_mnist_SVIGP_Hensman (:544-760): deep SVIGP with free variational
parameters, same loaders / inducing-point init, per-epoch train MSE and the conditional-generation MSE on the test set.
Other --elbo values (VAE, CVAE, GPVAE_Casale*) are baselines outside this build.
"""
import argparse
import json
import os
import pickle
import time

import numpy as np
import torch

from .SVGPVAE_model import _runtime, bacthing_predict_SVGPVAE_rotated_mnist, batching_encode_SVGPVAE, mnistSVGP
from .utils import batches, generate_init_inducing_points, import_rotated_mnist, parse_opt_regime
from .VAE_utils import mnistVAE


def build_parser():
    p = argparse.ArgumentParser(description="Rotated MNIST experiment (MI355X SVGPVAE_Hensman path).")
    p.add_argument('--expid', type=str, default="debug_MNIST")
    p.add_argument('--base_dir', type=str, default=os.getcwd())
    p.add_argument('--elbo', type=str, default="VAE",
                   choices=['VAE', 'CVAE', 'SVGPVAE_Hensman', 'SVGPVAE_Titsias', 'GPVAE_Casale', 'GPVAE_Casale_batch',
                            'SVIGP_Hensman'])
    p.add_argument('--mnist_data_path', type=str, default='MNIST data/')
    p.add_argument('--batch_size', type=int, default=256)
    p.add_argument('--nr_epochs', type=int, default=1000)
    p.add_argument('--beta', type=float, default=0.001)
    p.add_argument('--nr_inducing_points', type=float, default=2)
    p.add_argument('--save', action="store_true")
    p.add_argument('--GP_joint', action="store_true")
    p.add_argument('--ip_joint', action="store_true")
    p.add_argument('--ov_joint', action="store_true")
    p.add_argument('--lr', type=float, default=0.001)
    p.add_argument('--save_model_weights', action="store_true")
    p.add_argument('--dataset', type=str, choices=['3', '36', '13679'], default='3')
    p.add_argument('--show_pics', action="store_true")
    p.add_argument('--opt_regime', type=str, default=['joint-1000'], nargs="+")
    p.add_argument('--L', type=int, default=16)
    p.add_argument('--clip_qs', action="store_true")
    p.add_argument('--ram', type=float, default=1.0)
    p.add_argument('--test_set_metrics', action='store_true')
    p.add_argument('--GECO', action='store_true')
    p.add_argument('--alpha', type=float, default=0.99)
    p.add_argument('--kappa_squared', type=float, default=0.020)
    p.add_argument('--object_kernel_normalize', action='store_true')
    p.add_argument('--save_latents', action='store_true')
    p.add_argument('--jitter', type=float, default=0.000001)
    p.add_argument('--PCA', action="store_true")
    p.add_argument('--bias_analysis', action='store_true')
    p.add_argument('--M', type=int, default=8)
    # additions of this build
    p.add_argument('--train_file', type=str, default=None,
                   help="pickle used as the train set instead of train_data<dataset>.p (absent from the reference checkout)")
    p.add_argument('--eval_every', type=int, default=10, help="reference: every 10 epochs")
    p.add_argument('--seed', type=int, default=0, help="seed of the Keras-style weight init and numpy")
    p.add_argument('--epsilon_seed', type=int, default=None,
                   help="reproducible N(0,1) draws of SVGPVAE_model.py:901: batch i of epoch e uses "
                        "numpy.random.RandomState(1000 e + i + seed).randn(rows, L) instead of the on-device generator")
    p.add_argument('--split_grad_exchange', action='store_true',
                   help="data parallel: the gradient all-reduce in two parts, the first (decoder + GP parameters) beside the encoder's "
                        "reverse pass (cfg.split_grad_exchange; a fallback for slow small-message all-reduces)")
    p.add_argument('--log_json', type=str, default=None,
                   help="rank 0 writes the per-step log (elbo, recon_loss, C_ma, lagrange_mult), the evaluation series and the "
                        "final flat parameter vector to this file")
    return p


def run_experiment_rotated_mnist_SVGPVAE(args, args_dict=None, ctx=None):
    """MNIST_experiment.py:30-541 for elbo == SVGPVAE_Hensman.  Returns a dict of the logged series.

    Data parallel (svgp_vae_amd/dp.py): launched as `python -m torch.distributed.run --nproc-per-node G -m
    svgp_vae_amd.MNIST_experiment ...`, every rank takes a contiguous row range of every batch (incl. the ragged last one),
    c = N_train / b_global, the engines all-reduce statistics and gradients; rank 0 alone evaluates, prints and writes files."""
    from .dp import DistContext, attach_library_comm, run_sharded_epochs
    if args.elbo not in ("SVGPVAE_Hensman", "SVGPVAE_Titsias"):
        raise NotImplementedError(f"--elbo {args.elbo}: only SVGPVAE_Hensman / SVGPVAE_Titsias are built "
                                  f"(see DESIGN.md section 9)")
    ctx = (ctx or DistContext()).init()
    root = ctx.rank == 0
    np.random.seed(args.seed)                    # every rank: the same initial parameters (they are never broadcast)
    n = len(args.dataset)
    ending = args.dataset + ".p"
    train, ev, te, train_batches = import_rotated_mnist(args.mnist_data_path, ending, args.batch_size,
                                                        train_file=args.train_file)
    N_train = len(train["images"]) if args.train_file else n * 4050
    N_eval, N_test = len(ev["images"]), len(te["images"])

    chkpnt_dir = None
    if args.save and root:
        stamp = time.strftime("%d_%m_%Y__at__%H_%M_%S")
        chkpnt_dir = os.path.join(args.base_dir, args.expid, f"{args.elbo}_{args.beta}__on__{stamp}") + "/"
        os.makedirs(chkpnt_dir + "pics/", exist_ok=True)
        json.dump({k: v for k, v in (args_dict or vars(args)).items() if not callable(v)}, open(chkpnt_dir + "args.json", "wt"))

    # ---- model (MNIST_experiment.py:82-115)
    VAE = mnistVAE(L=args.L, seed=args.seed, device=ctx.device)
    inducing_points_init = generate_init_inducing_points(None, n=args.nr_inducing_points, remove_test_angle=None,
                                                         PCA=args.PCA, M=args.M, aux_data=train["aux_data"])
    ip_joint, GP_joint = not args.ip_joint, not args.GP_joint          # sic: passed as fixed_* (:96-98)
    if args.ov_joint:
        if args.PCA:
            object_vectors_init = pickle.load(open(args.mnist_data_path + f'pca_ov_init{args.dataset}.p', 'rb'))
        else:
            object_vectors_init = np.random.normal(0, 1.5, n * 400 * args.M).reshape(n * 400, args.M)
    else:
        object_vectors_init = None
    SVGP_ = mnistSVGP(titsias='Titsias' in args.elbo, fixed_inducing_points=ip_joint, initial_inducing_points=inducing_points_init,
                      fixed_gp_params=GP_joint, object_vectors_init=object_vectors_init, name='main',
                      jitter=args.jitter, N_train=N_train, L=args.L, K_obj_normalize=args.object_kernel_normalize,
                      device=ctx.device)
    kappa = float(np.sqrt(args.kappa_squared))
    # the cgen path runs the statistics over all train rows; the capacity is the SAME on every rank (the layout of the exchange
    # blocks is a function of it)
    b_cap = max(args.batch_size, N_train)
    rt = _runtime(VAE, SVGP_, args.clip_qs, args.GECO, kappa, b_cap, alpha_flag=args.alpha, lr=args.lr,
                  beta=args.beta, rank=ctx.rank, world_size=ctx.world,
                  split_grad_exchange=getattr(args, "split_grad_exchange", False))
    eng = rt.eng
    dev = eng.device
    comm_fallback = attach_library_comm(eng, ctx)
    if root:
        print(f"Number of train params: {eng.pl.n_total}")
        if ctx.multi:
            print(f"Data parallel over {ctx.world} ranks: " + ("RCCL communicator of the library, collectives on the compute stream"
                  if eng.comm is not None else f"torch.distributed all-reduces between the phases ({comm_fallback})"), flush=True)

    # ---- data resident in HBM; batches are copied device-to-device into fixed staging buffers so that
    #      one captured hipGraph per batch size serves every batch
    t64 = lambda a: torch.tensor(a, dtype=torch.float64, device=dev).contiguous()
    d_train_img, d_train_aux = t64(train["images"]), t64(train["aux_data"])
    d_eval_img, d_eval_aux = t64(ev["images"]), t64(ev["aux_data"])
    d_test_img, d_test_aux = t64(te["images"]), t64(te["aux_data"])
    stage_img = torch.zeros(args.batch_size, 28, 28, 1, dtype=torch.float64, device=dev)
    stage_aux = torch.zeros(args.batch_size, 2 + args.M, dtype=torch.float64, device=dev)
    # the N(0,1) draw of SVGPVAE_model.py:901 is made on device; `args.epsilon_fn(epoch, batch index, rows, L) -> array`
    # (not a CLI flag: set by callers that need a reproducible trajectory, tests/test_gpu_api.py) makes it an input
    eps_fn = getattr(args, "epsilon_fn", None)
    if eps_fn is None and getattr(args, "epsilon_seed", None) is not None:
        eps_fn = lambda epoch, i, rows, L_: np.random.RandomState(1000 * epoch + i + args.epsilon_seed).randn(rows, L_)
    stage_eps = torch.zeros(args.batch_size, args.L, dtype=torch.float64, device=dev) if eps_fn else None
    eng.stream.wait_stream(torch.cuda.current_stream(dev))   # uploads / zero-fills above ran on torch's stream
    graphs = {}

    def local_step(llo, lhi, lo, hi, epoch, i):
        b, b_global = lhi - llo, hi - lo
        with torch.cuda.stream(eng.stream):
            stage_img[:b].copy_(d_train_img[llo:lhi])
            stage_aux[:b].copy_(d_train_aux[llo:lhi])
            if eps_fn:                 # the draw of the GLOBAL batch; this rank's rows of it
                e = np.asarray(eps_fn(epoch, i, b_global, args.L))[llo - lo:lhi - lo]
                stage_eps[:b].copy_(torch.as_tensor(e, dtype=torch.float64), non_blocking=False)
        if ctx.multi:                  # collectives are not captured: the eager step (as fast as the replay)
            eng.set_batch_size(b, b_global)
            eng.bind(stage_img[:b], stage_aux[:b], stage_eps[:b] if eps_fn else None)
            eng.run(adam=True)
        else:
            if b not in graphs:
                eng.set_batch_size(b)
                eng.bind(stage_img[:b], stage_aux[:b], stage_eps[:b] if eps_fn else None)   # None: drawn on device
                eng.capture(("train", b), adam=True)
                graphs[b] = True
            eng.replay(("train", b))
        # the reference fetches elbo / recon_loss / C_ma / lagrange_mult every step (:334-340); one 128-byte read-back
        eng.synchronize()
        return eng.scalars()

    def on_epoch_end(epoch, log):
        if not ((epoch + 1) % args.eval_every == 0 or epoch + 1 == nr_epochs):
            return
        if root:       # parameters are replicated: rank 0 evaluates alone (collective-free stage calls), the others wait
            # eval-set reconstruction through the posterior (forward only, no optimiser step)
            ev_losses = []
            for lo, hi in batches(N_eval, args.batch_size):
                eng.set_batch_size(hi - lo)
                eng.bind(d_eval_img[lo:hi].contiguous(), d_eval_aux[lo:hi].contiguous(), None)
                with torch.cuda.stream(eng.stream):
                    eng.phase(0); eng.phase(1)
                eng.synchronize()
                rec = eng.ws_view("recon", (hi - lo, 28, 28, 1))
                ev_losses.append(float(torch.sum((d_eval_img[lo:hi] - rec) ** 2)) / 784.0)
            eval_mse = float(np.sum(ev_losses) / N_eval)
            # conditional generation on the test set (:457-486)
            means, vars_ = [], []
            for lo, hi in train_batches:
                mu, var, _ = batching_encode_SVGPVAE((d_train_img[lo:hi], d_train_aux[lo:hi]), VAE,
                                                     clipping_qs=args.clip_qs)
                means.append(mu); vars_.append(var)
            means, vars_ = torch.cat(means), torch.cat(vars_)
            cg = []
            for lo, hi in batches(N_test, args.batch_size):
                _, loss_ = bacthing_predict_SVGPVAE_rotated_mnist((d_test_img[lo:hi], d_test_aux[lo:hi]), VAE, SVGP_,
                                                                  means, vars_, d_train_aux)
                cg.append(float(loss_))
            cgen_mse = float(np.sum(cg) / N_test)
            log.setdefault("eval_mse", []).append((epoch, eval_mse)); log.setdefault("cgen_mse", []).append((epoch, cgen_mse))
            print(f"  eval recon MSE/px {eval_mse:.6f}   cgen test MSE/px {cgen_mse:.6f}   "
                  f"l_GP {float(SVGP_.l_GP):.4f} amplitude {float(SVGP_.amplitude):.4f}", flush=True)
            if chkpnt_dir:
                with open(chkpnt_dir + "pics/test_metrics.txt", "a") as f:
                    f.write(f"{epoch},{eval_mse},{cgen_mse}\n")
                if args.save_model_weights:
                    torch.save({"theta": eng.theta.cpu(), "adam_m": eng.adam_m.cpu(), "adam_v": eng.adam_v.cpu(),
                                "state": eng.state.cpu()}, chkpnt_dir + f"model_{epoch}.pt")
        ctx.barrier()

    nr_epochs, training_regime = parse_opt_regime(args.opt_regime)
    start = time.time()

    log = run_sharded_epochs(ctx, train_batches, nr_epochs, local_step, N_train=N_train, on_epoch_end=on_epoch_end)
    log.setdefault("eval_mse", []); log.setdefault("cgen_mse", [])
    log["world_size"], log["rank"] = ctx.world, ctx.rank
    log["rccl_ranks"] = eng.comm.world_size if eng.comm is not None else 0
    log["total_time"] = time.time() - start
    if root and getattr(args, "log_json", None):
        eng.synchronize()
        json.dump(dict({k: v for k, v in log.items()}, theta=eng.theta.cpu().tolist(), adam_t=eng.scalars()["adam_t"],
                       param_order=list(eng.shapes)), open(args.log_json, "wt"))
    log["_engine"] = eng                     # (the final parameters / optimiser state for callers; not serialised)
    return log


def run_experiment_rotated_mnist_SVIGP_Hensman(args, args_dict=None):
    """MNIST_experiment.py:544-760.  Returns a dict of the logged series."""
    from .SVIGP_Hensman_model import SVIGP_Hensman, SVIGP_Hensman_decoder, SvigpStepEngine
    np.random.seed(args.seed)
    n = len(args.dataset)
    ending = args.dataset + ".p"
    train, ev, te, train_batches = import_rotated_mnist(args.mnist_data_path, ending, args.batch_size,
                                                        train_file=args.train_file)
    N_train = len(train["images"]) if args.train_file else n * 4050
    N_test = len(te["images"])
    chkpnt_dir = None
    if args.save:
        stamp = time.strftime("%d_%m_%Y__at__%H_%M_%S")
        chkpnt_dir = os.path.join(args.base_dir, args.expid, f"{args.elbo}_{args.beta}__on__{stamp}") + "/"
        os.makedirs(chkpnt_dir + "pics/", exist_ok=True)
        json.dump(args_dict or vars(args), open(chkpnt_dir + "args.json", "wt"))
    VAE = SVIGP_Hensman_decoder(L=args.L, seed=args.seed)                                  # :591
    inducing_points_init = generate_init_inducing_points(None, n=args.nr_inducing_points, remove_test_angle=None,
                                                         PCA=args.PCA, M=args.M, aux_data=train["aux_data"])
    if args.ov_joint:                                                                        # :603-612
        if args.PCA:
            object_vectors_init = pickle.load(open(args.mnist_data_path + f'pca_ov_init{args.dataset}.p', 'rb'))
        else:
            object_vectors_init = np.random.normal(0, 1.5, n * 400 * args.M).reshape(n * 400, args.M)
    else:
        object_vectors_init = None
    SVGP_ = SVIGP_Hensman(fixed_inducing_points=not args.ip_joint, initial_inducing_points=inducing_points_init,
                          fixed_gp_params=not args.GP_joint, object_vectors_init=object_vectors_init, name='main',
                          jitter=args.jitter, N_train=N_train, L=args.L, K_obj_normalize=args.object_kernel_normalize,
                          dtype=np.float64)                                                  # :616-620
    eng = SvigpStepEngine(VAE, SVGP_, b_max=args.batch_size, lr=args.lr)
    dev = eng.dev
    t64 = lambda a: torch.tensor(a, dtype=torch.float64, device=dev).contiguous()
    d_train_img, d_train_aux = t64(train["images"]), t64(train["aux_data"])
    d_test_img, d_test_aux = t64(te["images"]), t64(te["aux_data"])
    nr_epochs = args.nr_epochs if args.opt_regime == ['joint-1000'] else parse_opt_regime(args.opt_regime)[0]
    log = dict(epoch=[], elbo=[], recon_loss=[], cgen_mse=[], epoch_time=[])
    start = time.time()
    for epoch in range(nr_epochs):
        t0 = time.time()
        elbos, losses = [], []
        for lo, hi in train_batches:                                                         # :700-706
            eng.step(d_train_img[lo:hi], d_train_aux[lo:hi], adam=True)
            sc = eng.scalars()
            elbos.append(sc["elbo"]); losses.append(sc["recon_loss"])
        mse = float(np.sum(losses) / N_train)
        log["epoch"].append(epoch); log["elbo"].append(float(np.mean(elbos))); log["recon_loss"].append(mse)
        log["epoch_time"].append(time.time() - t0)
        if (epoch + 1) % args.eval_every == 0 or epoch + 1 == nr_epochs:
            print(f"Epoch {epoch}, mean ELBO per batch: {np.mean(elbos)}")
            print(f"MSE loss on train set for epoch {epoch} : {mse}")
            cg = [float(eng.predict(d_test_img[lo:hi], d_test_aux[lo:hi])[1]) for lo, hi in batches(N_test, args.batch_size)]
            cgen = float(np.sum(cg) / N_test)                                               # :733-748
            log["cgen_mse"].append((epoch, cgen))
            l_GP, amp, _, _ = SVGP_.variable_summary()
            print(f"Conditional generation MSE loss on test set for epoch {epoch}: {cgen}   "
                  f"(l_GP {float(l_GP):.4f}, amplitude {float(amp):.4f}, noise {float(eng.vp['noise'][0]):.4f})", flush=True)
            if chkpnt_dir:
                with open(chkpnt_dir + "pics/test_metrics.txt", "a") as f:
                    f.write(f"{epoch},{mse},{cgen}\n")
                if args.save_model_weights:
                    torch.save({"theta": eng.mn.theta.cpu(), "phi": eng.phi.cpu(), "state": eng.mn.state.cpu()},
                               chkpnt_dir + f"model_{epoch}.pt")
    log["total_time"] = time.time() - start
    return log


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.elbo in ("SVGPVAE_Hensman", "SVGPVAE_Titsias"):
        from .dp import DistContext
        ctx = DistContext()
        try:
            return run_experiment_rotated_mnist_SVGPVAE(args, vars(args), ctx)
        finally:
            ctx.close()
    if args.elbo == "SVIGP_Hensman":
        return run_experiment_rotated_mnist_SVIGP_Hensman(args, vars(args))
    raise NotImplementedError(f"--elbo {args.elbo}: only SVGPVAE_Hensman / SVGPVAE_Titsias / SVIGP_Hensman are built "
                              f"(see DESIGN.md section 9)")


if __name__ == "__main__":
    main()
