"""Moving-ball experiment driver with the reference's CLI (BALL_experiment.py:283-337 flags and defaults), running
every step on the HIP library:

    python -m svgp_vae_amd.BALL_experiment --elbo VAE                               # BASELINE configs[0]
    python -m svgp_vae_amd.BALL_experiment --elbo SVGPVAE_Hensman --clip_qs --m 15 --GP_joint --ip_joint

Mirrors run_experiment (BALL_experiment.py:21-280): data-synthesis settings (:37-48: 35 videos of tmax 32x32 frames,
ball radius 3; model_lt = 0.001 for --elbo VAE), the fixed test batches (Make_Video_batch seeds 0..9, cached in
<base_dir>/Test_Batches_<vidlt>_<tmax>.pkl), a fresh synthetic training batch per step (build_video_batch_graph), the
SVGP objects with the inverted `fixed_*` flags (:100-103), loss = -mean(elbo) with TF1 Adam and the optional gradient
clip (:116-136), and every `--eval_every` (reference: 1000) steps the test-batch diagnostics at beta = 1 (:219-247):
ELBO terms, posterior ranges, MSE of the affinely rotated latent paths (MSE_rotation), appended to
<chkpnt>/res/ELBO_log.jsonl when --save.  Plotting / pandas logging / TF checkpoints are not reproduced; --save_model
writes the flat parameter vector + Adam moments (BALL_experiment.py:270-272 saves every 50000 steps).
"""
import argparse
import json
import os
import pickle
import time

import numpy as np
import torch

from . import ball


def build_parser():
    default_base_dir = os.getcwd()
    p = argparse.ArgumentParser(description='Moving ball experiment')
    p.add_argument('--steps', type=int, default=25000, help='Number of steps of Adam')
    p.add_argument('--beta0', type=float, default=1, help='initial beta annealing value')
    p.add_argument('--elbo', type=str, choices=['GPVAE_Pearce', 'VAE', 'NP', 'SVGPVAE_Hensman', 'SVGPVAE_Titsias'],
                   default='GPVAE_Pearce', help='Structured Inf Nets ELBO or Neural Processes ELBO')
    p.add_argument('--modellt', type=float, default=2, help='time scale of model to fit to data')
    p.add_argument('--base_dir', type=str, default=default_base_dir, help='folder within a new dir is made for each run')
    p.add_argument('--expid', type=str, default="debug", help='give this experiment a name')
    p.add_argument('--ram', type=float, default=0.5, help='(ignored) fraction of GPU ram to use')
    p.add_argument('--seed', type=int, default=None, help='seed for rng')
    p.add_argument('--tmax', type=int, default=30, help='length of videos')
    p.add_argument('--m', type=int, default=15, help='number of inducing points')
    p.add_argument('--GP_joint', action="store_true", help='GP hyperparams joint optimization.')
    p.add_argument('--ip_joint', action="store_true", help='Inducing points joint optimization.')
    p.add_argument('--clip_qs', action="store_true", help='Clip variance of inference network.')
    p.add_argument('--show_pics', action="store_true", help='(ignored) Show images during training.')
    p.add_argument('--save', action="store_true", help='Save model metrics.')
    p.add_argument('--squares_circles', action="store_true", help='(ignored) plot squares and circles.')
    p.add_argument('--ip_min', type=int, default=1, help='ip start')
    p.add_argument('--ip_max', type=int, default=30, help='ip end')
    p.add_argument('--jitter', type=float, default=1e-9, help='noise for GP operations (inverse, cholesky)')
    p.add_argument('--clip_grad', action="store_true", help='Whether or not to clip gradients.')
    p.add_argument('--vidlt', type=float, default=2, help='time scale for data generation')
    p.add_argument('--GP_init', type=float, default=2,
                   help='Initial value for GP kernel length scale. Used when running --GP_joint .')
    # additions of this build
    p.add_argument('--eval_every', type=int, default=1000, help='diagnostics period in steps (reference: 1000)')
    p.add_argument('--hidden', type=int, default=500, help='width of the MLP hidden layers (reference: 500)')
    p.add_argument('--save_model', action="store_true", help='write parameters + Adam state at the end')
    p.add_argument('--device', type=str, default="cuda:0")
    return p


def make_checkpoint_folder(base_dir, expid, extra=""):
    """utils.py:377-410 in spirit: <base_dir>/<expid>_<extra>_<timestamp>/ with res/ and pics/ inside."""
    d = os.path.join(base_dir, f"{expid}_{extra}_{time.strftime('%m_%d_%H_%M_%S')}") + "/"
    for sub in ("", "res", "pics", "preds"):
        os.makedirs(d + sub, exist_ok=True)
    return d


def load_or_make_test_batches(base_dir, vid_lt, tmax, px, py, batch, r):
    """BALL_experiment.py:54-62."""
    path = os.path.join(base_dir, "Test_Batches_{}_{}.pkl".format(vid_lt, tmax))
    if os.path.isfile(path):
        with open(path, "rb") as f:
            return pickle.load(f)
    tb = [ball.Make_Video_batch(tmax=tmax, px=px, py=py, lt=vid_lt, batch=batch, seed=s, r=r) for s in range(10)]
    with open(path, "wb") as f:
        pickle.dump(tb, f)
    return tb


def build_engine(args, batch=35, px=32, py=32):
    """The model of BALL_experiment.py:86-114 as a step engine."""
    tmax, seed = args.tmax, 0 if args.seed is None else args.seed
    common = dict(batch=batch, tmax=tmax, px=px, py=py, hidden=args.hidden, beta=args.beta0, clip_grad=args.clip_grad,
                  device=args.device, seed=seed)
    if args.elbo in ('GPVAE_Pearce', 'VAE', 'NP'):
        model_lt = 0.001 if args.elbo == 'VAE' else args.modellt
        return ball.PearceStepEngine(args.elbo, model_lt, 0.5, args.GP_joint, args.GP_init, **common)
    titsias = 'Titsias' in args.elbo
    mk = lambda name: ball.SVGP(titsias=titsias, num_inducing_points=args.m, fixed_inducing_points=not args.ip_joint,
                                tmin=1, tmax=tmax, vidlt=args.vidlt, fixed_gp_params=not args.GP_joint, name=name,
                                jitter=args.jitter, ip_min=args.ip_min, ip_max=args.ip_max, GP_init=args.GP_init)
    return ball.BallStepEngine(mk('x'), mk('y'), clip_qs=args.clip_qs, **common)


def evaluate(eng, TT, TD, beta0):
    """Test-batch diagnostics at beta = 1 (BALL_experiment.py:219-247); restores beta0 afterwards."""
    eng.set_scalars(beta=1.0)
    vid = torch.as_tensor(np.asarray(TD), dtype=torch.float64, device=eng.dev)
    eng.step(vid, None, adam=False, backward=False)
    out = eng.outputs()
    sc = eng.scalars()
    svgp = isinstance(eng, ball.BallStepEngine)
    p_m, p_v, q_m, q_v = (out[5], out[6], out[7], out[8]) if svgp else (out[3], out[4], out[5], out[6])
    p_m, p_v, q_m, q_v = [t.cpu().numpy() for t in (p_m, p_v, q_m, q_v)]
    _, _, MSE, _ = ball.MSE_rotation(p_m, np.asarray(TT), p_v)
    res = {"elbo": sc["elbo"], "recon": sc["recon_loss"], "prior_kl": sc["kl_term"], "MSE": float(MSE),
           "min qs_var": float(q_v.min()), "max qs_var": float(q_v.max()), "min q_var": float(p_v.min()),
           "max q_var": float(p_v.max()), "min qs_mean": float(q_m.min()), "max qs_mean": float(q_m.max()),
           "min q_mean": float(p_m.min()), "max q_mean": float(p_m.max()),
           "l_GP_x": float(eng.params["l_x"][0]), "l_GP_y": float(eng.params["l_y"][0])}
    if svgp:
        res.update({"SVGP elbo": sc["inside_elbo"], "ce term": sc["ce_term"], "SVGP elbo recon": sc["inside_recon"],
                    "SVGP elbo KL": sc["inside_kl"], "inducing_points_x": eng.params["ip_x"].cpu().tolist(),
                    "inducing_points_y": eng.params["ip_y"].cpu().tolist()})
    eng.set_scalars(beta=beta0)
    return res


def run_experiment(args):
    batch, px, py, r = 35, 32, 32, 3                                   # BALL_experiment.py:37-43
    model_lt = 0.001 if args.elbo == 'VAE' else args.modellt
    assert model_lt == args.vidlt or args.GP_joint or args.elbo == 'VAE', \
        "GP params of data and model should match. Except when doing a joint optimization of GP parameters or when " \
        "fitting normal VAE."
    chk = None
    if args.save or args.save_model:
        chk = make_checkpoint_folder(args.base_dir, args.expid, args.elbo + "_" + str(args.beta0))
        print("\nCheckpoint Directory:\n" + str(chk) + "\n")
    Test_Batches = load_or_make_test_batches(args.base_dir, args.vidlt, args.tmax, px, py, batch, r)
    eng = build_engine(args, batch, px, py)
    src = ball.VideoBatchSource(tmax=args.tmax, px=px, py=py, lt=args.vidlt, batch=batch,
                                seed=1 if args.seed is None else args.seed, r=r, device=args.device)
    print("\n\nTrainable variables:")
    for k, s in eng.shapes.items():
        print(" ", k, tuple(s))
    log, t0 = [], time.time()
    for t in range(args.steps):
        vid = src()                                   # torch's current stream; eng.step waits for it
        vid.record_stream(eng.stream)
        eng.step(vid, None, adam=True)
        g_s = t + 1
        if g_s % args.eval_every == 0 or g_s == args.steps:
            TT, TD = Test_Batches[0]
            res = evaluate(eng, TT, TD, args.beta0)
            res.update(Step=g_s, Beta=args.beta0, Time=time.time() - t0)
            print(str(g_s) + ": elbo " + str(res["elbo"]))
            print("Recon term: {}. KL term: {}.".format(res["recon"], res["prior_kl"]))
            if 'SVGPVAE' in args.elbo:
                print("L{} elbo term: {}. CE term: {}.".format(2 if 'Titsias' in args.elbo else 3, res["SVGP elbo"],
                                                             res["ce term"]))
            print("VAE posterior variance range: min {}, max  {}".format(res["min qs_var"], res["max qs_var"]))
            print("GP approx posterior variance range: min {}, max {}".format(res["min q_var"], res["max q_var"]))
            print('MSE : {}'.format(res["MSE"]))
            log.append(res)
            if args.save:
                with open(chk + "res/ELBO_log.jsonl", "a") as f:
                    f.write(json.dumps(res) + "\n")
    if args.save_model and chk:
        eng.stream.synchronize()
        torch.save({"theta": eng.theta.cpu(), "adam_m": eng.adam_m.cpu(), "adam_v": eng.adam_v.cpu(),
                    "state": eng.state.cpu(), "shapes": eng.shapes}, chk + "model.pt")
        print("\n\nModel Saved: " + chk + "\n\n")
    return log


def main(argv=None):
    args = build_parser().parse_args(argv)
    return run_experiment(args)


if __name__ == "__main__":
    main()
