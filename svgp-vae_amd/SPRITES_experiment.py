"""`python -m svgp_vae_amd.SPRITES_experiment`: counterpart of the reference's SPRITES_experiment.py (its flags,
its epoch loop :376-455, its test pipeline :470-560) on the MI355X engine.

Differences that are stated, not hidden:
  * data files.  The reference reads TFRecords written by its own SPRITES_utils.save_sprites (TensorFlow needed to
    parse them).  Here `--sprites_data_path` holds `train.npz` and `test_character.npz`, each with
    frames (N,64,64,3), char_IDs (N), action_IDs (N), rows ordered by character as in the TFRecords
    (N_frames_per_character consecutive rows per character; test characters: all N_actions frames).
    `--synthetic n_train_chars,n_test_chars` generates random data of that layout instead.
  * `--repr_nn_pretrain yes_fixed|yes_joint` (classification pre-training of the representation network,
    :139-151,325-357) runs `sprites.pretrain_repr_NN` on the train frames (classes = char_IDs); the default here is 'no'.
  * `--elbo VAE`, pandas / matplotlib logging, `--show_pics`, `--ram` are accepted and ignored or rejected with a message.
Only SVGPVAE_Hensman / SVGPVAE_Titsias are built.  float64 (the reference uses float32)."""
import argparse
import json
import os
import time

import numpy as np
import torch

from . import sprites as S
from .utils import parse_opt_regime

N_FRAMES_PER_CHARACTER_TRAIN = 50     # SPRITES_experiment.py:36


def build_parser():
    p = argparse.ArgumentParser(description='Train SVGPVAE for SPRITES data.')
    p.add_argument('--expid', type=str, default="debug_SPRITES")
    p.add_argument('--base_dir', type=str, default=os.getcwd())
    p.add_argument('--elbo', type=str, choices=['VAE', 'SVGPVAE_Hensman', 'SVGPVAE_Titsias'], default='VAE')
    p.add_argument('--sprites_data_path', type=str, default='SPRITES_data/')
    p.add_argument('--batch_size', type=int, default=500)
    p.add_argument('--nr_epochs', type=int, default=50)
    p.add_argument('--beta', type=float, default=0.001)
    p.add_argument('--m', type=int, default=1)
    p.add_argument('--save', action="store_true")
    p.add_argument('--ip_joint', action="store_true")
    p.add_argument('--GPLVM_joint', action="store_true")
    p.add_argument('--lr', type=float, default=0.001)
    p.add_argument('--save_model_weights', action="store_true")
    p.add_argument('--show_pics', action="store_true")
    p.add_argument('--beta_schedule_switch', type=int, default=100)
    p.add_argument('--opt_regime', type=str, default=['joint-50'], nargs="+")
    p.add_argument('--L', type=int, default=64)
    p.add_argument('--L_action', type=int, default=8)
    p.add_argument('--L_character', type=int, default=16)
    p.add_argument('--clip_qs', action="store_true")
    p.add_argument('--ram', type=float, default=1.0)
    p.add_argument('--GECO', action='store_true')
    p.add_argument('--alpha', type=float, default=0.99)
    p.add_argument('--kappa_squared', type=float, default=0.0075)
    p.add_argument('--jitter', type=float, default=0.01)
    p.add_argument('--PCA', action="store_true")
    p.add_argument('--N_context', type=int, default=36)
    p.add_argument('--test_set_metrics', action='store_true')
    p.add_argument('--clip_grad', action="store_true")
    p.add_argument('--repr_nn_pretrain', type=str, choices=['no', 'yes_fixed', 'yes_joint'], default='no')
    p.add_argument('--lr_repr_nn', type=float, default=0.01)
    p.add_argument('--nr_epochs_repr_nn', type=int, default=400)
    p.add_argument('--batch_size_repr_nn', type=int, default=5000)
    p.add_argument('--object_kernel_normalize', action='store_true')
    p.add_argument('--K_SE', action='store_true')
    p.add_argument('--GP_joint', action="store_true")
    p.add_argument('--clip_grad_thres', type=float, default=1000000.0)
    # additions of this build
    p.add_argument('--synthetic', type=str, default=None,
                   help="n_train_chars,n_test_chars: random frames in the file layout instead of --sprites_data_path")
    p.add_argument('--N_actions', type=int, default=72, help="frames per test character (72 in the dataset)")
    p.add_argument('--frames_per_character', type=int, default=N_FRAMES_PER_CHARACTER_TRAIN)
    p.add_argument('--batch_size_test_char', type=int, default=576)
    p.add_argument('--eval_every', type=int, default=5, help="reference: every 5 epochs")
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--epsilon_seed', type=int, default=None,
                   help="reproducible N(0,1) draws: batch i of epoch e uses numpy.random.RandomState(1000 e + i + seed)"
                        ".randn(rows, L) instead of the on-device generator")
    p.add_argument('--log_json', type=str, default=None, help="rank 0 writes the per-step log and the final parameters here")
    return p


def _load(args):
    if args.synthetic:
        n_tr, n_te = (int(v) for v in args.synthetic.split(","))
        rs = np.random.RandomState(args.seed)
        fpc, na = args.frames_per_character, args.N_actions
        def make(n_char, per):
            base = rs.rand(n_char, 1, 8, 8, 3).repeat(8, 2).repeat(8, 3)                    # a "character" texture
            act = rs.rand(1, na, 8, 8, 3).repeat(8, 2).repeat(8, 3)                         # an "action" texture
            aid = np.stack([rs.permutation(na)[:per] for _ in range(n_char)])               # (n_char, per)
            fr = 0.6 * base + 0.4 * act[0][aid] + 0.02 * rs.randn(n_char, per, 64, 64, 3)
            return dict(frames=np.clip(fr, 0, 1).reshape(-1, 64, 64, 3), char_IDs=np.repeat(np.arange(n_char), per),
                        action_IDs=aid.reshape(-1))
        te = make(n_te, na)
        te["action_IDs"] = np.tile(np.arange(na), n_te)            # test characters: every action once, in order
        return make(n_tr, fpc), te
    out = []
    for name in ("train.npz", "test_character.npz"):
        f = os.path.join(args.sprites_data_path, name)
        if not os.path.exists(f):
            raise FileNotFoundError(f"{f}: expected npz with frames (N,64,64,3), char_IDs (N), action_IDs (N) "
                                    f"(see the module docstring), or pass --synthetic")
        d = np.load(f)
        out.append({k: d[k] for k in ("frames", "char_IDs", "action_IDs")})
    return out


def run_experiment_sprites_SVGPVAE(args, dict_=None, ctx=None):
    """SPRITES_experiment.py:30-560.  Data parallel (svgp_vae_amd/dp.py): under `python -m torch.distributed.run --nproc-per-node G
    -m svgp_vae_amd.SPRITES_experiment ...` every rank takes whole `frames_per_character` groups of every batch (the GP posterior
    of a batch couples all its frames through the all-reduced statistics; a character's frames share their representation
    vector, so groups are never cut), c = N_train / b_global; rank 0 alone evaluates, prints and writes files."""
    from .dp import DistContext, make_comm, shard_batch
    if "SVGPVAE" not in args.elbo:
        raise NotImplementedError(f"--elbo {args.elbo}: only SVGPVAE_Hensman / SVGPVAE_Titsias are built")
    assert np.sum([args.object_kernel_normalize, args.K_SE]) <= 1, \
        "At most one of GP kernel engineering flags can be used at once!"                      # :43-44
    fpc, N_actions = args.frames_per_character, args.N_actions
    assert args.batch_size % fpc == 0, f"Batch size needs to be divisible by {fpc}"         # :40-41
    assert args.batch_size_test_char % N_actions == 0 and 0 < args.N_context < N_actions
    ctx = (ctx or DistContext()).init()
    root = ctx.rank == 0
    say = print if root else (lambda *a, **k: None)
    assert args.batch_size // fpc >= ctx.world, \
        f"{ctx.world} ranks need at least {ctx.world} character groups per batch (batch_size >= {ctx.world * fpc})"
    np.random.seed(args.seed)                    # every rank: the same data / initial parameters (they are never broadcast)
    train, test = _load(args)
    N_train, N_test = len(train["frames"]), len(test["frames"])
    assert N_train % args.batch_size == 0 or N_train > args.batch_size, "need at least one full train batch"
    chkpnt_dir = None
    if args.save and root:
        stamp = time.strftime("%d_%m_%Y__at__%H_%M_%S")
        chkpnt_dir = os.path.join(args.base_dir, args.expid, f"{args.elbo}_{args.beta}__on__{stamp}") + "/"
        os.makedirs(chkpnt_dir + "pics/", exist_ok=True)
        json.dump(dict_ or vars(args), open(chkpnt_dir + "args.json", "wt"))

    # ---- model (SPRITES_experiment.py:82-121)
    if args.PCA:        # :96-99 (the reference reads sprites_train_dict.p; here: the loaded / synthetic training set)
        from .SPRITES_utils import sprites_PCA_init
        GPLVM_init, IP_init = sprites_PCA_init(
            dict(frames=train["frames"], aux_data=np.stack([train["char_IDs"], train["action_IDs"]], 1)), m=args.m,
            L_action=args.L_action, L_character=args.L_character, N_action=N_actions)
    else:
        GPLVM_init = np.random.normal(0, 1.5, N_actions * args.L_action).reshape(N_actions, args.L_action)
        IP_init = np.random.normal(0, 1.5, N_actions * args.m * (args.L_action + args.L_character)) \
            .reshape(N_actions * args.m, args.L_action + args.L_character)
    device = ctx.device
    VAE = S.spritesVAE(L=args.L, seed=args.seed)
    repr_NN = S.sprites_representation_network(L=args.L_character)
    SVGP_ = S.spritesSVGP(titsias='Titsias' in args.elbo, fixed_inducing_points=not args.ip_joint,
                          initial_inducing_points=IP_init, name='main', jitter=args.jitter, N_train=N_train,
                          L_action=args.L_action, initial_GPLVM_action=GPLVM_init, L_character=args.L_character,
                          fixed_GPLVM=not args.GPLVM_joint, K_obj_normalize=args.object_kernel_normalize, L=args.L,
                          K_SE=args.K_SE, fixed_GP_params=not args.GP_joint)
    b_max = max(args.batch_size, args.batch_size_test_char)      # the same capacity on every rank (layout of the exchange blocks)
    comm, comm_fallback = make_comm(ctx, device)
    eng = S.SpritesStepEngine(VAE, repr_NN, SVGP_, b_max=b_max, seg_len=fpc, clip_qs=args.clip_qs, geco=args.GECO,
                              kappa_squared=args.kappa_squared, alpha=args.alpha, beta=args.beta, lr=args.lr,
                              clip_grad=args.clip_grad_thres if args.clip_grad else None, device=device,
                              rank=ctx.rank, world_size=ctx.world, comm=comm)
    S._attach(eng, SVGP_, VAE, repr_NN)
    dev = eng.dev
    say(f"Number of train params: {eng.theta.numel()}")
    if ctx.multi:
        say(f"Data parallel over {ctx.world} ranks: " + ("RCCL communicator of the library, collectives on the compute stream"
            if comm_fallback is None else f"torch.distributed collectives ({comm_fallback})"), flush=True)
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=dev).contiguous()
    d_tr, a_tr = t64(train["frames"]), t64(train["action_IDs"])
    d_te, a_te = t64(test["frames"]), t64(test["action_IDs"])

    train_seg, train_rep = S.aux_data_sprites_utils(args.batch_size, fpc, fpc)
    bt = args.batch_size_test_char
    cgen_seg, cgen_rep = S.aux_data_sprites_utils(int(bt * args.N_context / N_actions), args.N_context,
                                                  N_actions - args.N_context)                # :370-372
    nr_epochs, training_regime = parse_opt_regime(args.opt_regime)
    log = dict(elbo=[], recon_loss=[], recon_mse_test=[], cgen_mse=[], steps=[])
    eps_fn = getattr(args, "epsilon_fn", None)           # (not a CLI flag: reproducible trajectories for tests)
    if eps_fn is None and getattr(args, "epsilon_seed", None) is not None:
        eps_fn = lambda epoch, i, rows, L_: np.random.RandomState(1000 * epoch + i + args.epsilon_seed).randn(rows, L_)
    if 'yes' in args.repr_nn_pretrain:                                     # :325-357
        bs = min(args.batch_size_repr_nn, N_train)
        log["repr_pretrain"] = S.pretrain_repr_NN(eng, d_tr, t64(train["char_IDs"]), nr_epochs=args.nr_epochs_repr_nn,
                                                  lr=args.lr_repr_nn, batch_size=bs,
                                                  n_classes=max(1000, int(train["char_IDs"].max()) + 1), seed=args.seed,
                                                  carry_slots='fixed' not in args.repr_nn_pretrain)
        eng.freeze_repr = 'fixed' in args.repr_nn_pretrain
    first_step = True
    start = time.time()
    for epoch in range(nr_epochs):
        elbos, losses = [], []
        for i, lo in enumerate(range(0, N_train - args.batch_size + 1, args.batch_size)):
            if args.GECO and first_step:                    # :381-385: alpha = 0 on the very first GECO step
                eng.set_scalars(alpha=0.0)
            llo, lhi = shard_batch(lo, lo + args.batch_size, ctx.world, ctx.rank, fpc)      # whole character groups
            eps = None
            if eps_fn is not None:                          # the draw of the GLOBAL batch; this rank's rows of it
                eps = t64(np.asarray(eps_fn(epoch, i, args.batch_size, args.L))[llo - lo:lhi - lo])
            eng.step(d_tr[llo:lhi], a_tr[llo:lhi], eps, adam=True, b_global=args.batch_size)
            sc = eng.scalars()                              # global values, identical on every rank
            elbos.append(sc["elbo"]); losses.append(sc["recon_loss"])
            log["steps"].append(dict(epoch=epoch, rows=args.batch_size, local_rows=lhi - llo, elbo=sc["elbo"],
                                     recon_loss=sc["recon_loss"], C_ma=sc["c_ma"], lagrange_mult=sc["lagrange"]))
            first_step = False
        # GECO recon_loss is sum_i(mean_pix - kappa^2); the reference prints it summed / N_train all the same (:432)
        log["elbo"].append(float(np.mean(elbos))); log["recon_loss"].append(float(np.sum(losses) / N_train))
        say(f"Epoch {epoch}, opt regime {training_regime[epoch]}, mean ELBO per batch: {log['elbo'][-1]}")
        say(f"MSE loss on train set for epoch {epoch} : {log['recon_loss'][-1]}", flush=True)
        if (epoch + 1) % args.eval_every:
            continue
        if not root:        # parameters are replicated: rank 0 evaluates alone (collective-free calls), the others wait
            ctx.barrier()
            continue
        # ---- 7.3.1 reconstruction of the test characters (:472-487): encode -> GP posterior of the same batch -> decode
        mse = []
        for lo in range(0, N_test - bt + 1, bt):
            out = S.forward_pass_SVGPVAE((d_te[lo:lo + bt], a_te[lo:lo + bt]), args.beta, VAE, SVGP_, 0.0, 1.0, args.alpha,
                                         np.sqrt(args.kappa_squared), clipping_qs=args.clip_qs, GECO=False,
                                         repr_NN=repr_NN, repeats=[N_actions], engine=_eval_engine(eng, N_actions))
            mse.append(float(torch.sum((d_te[lo:lo + bt] - out[9]) ** 2)) / (64 * 64 * 3))
        n_eval = (N_test // bt) * bt
        log["recon_mse_test"].append((epoch, float(np.sum(mse) / n_eval)))
        print(f"MSE loss on test set for epoch {epoch} : {log['recon_mse_test'][-1][1]}")
        # ---- 7.3.2 conditional generation (:499-545)
        n_full = (N_train // args.batch_size) * args.batch_size
        mu, var, aux = [], [], []
        for lo in range(0, n_full, args.batch_size):
            m_, v_, a_ = S.batching_encode_SVGPVAE((d_tr[lo:lo + args.batch_size], a_tr[lo:lo + args.batch_size]), VAE,
                                                   clipping_qs=args.clip_qs, repr_nn=repr_NN, segment_ids=train_seg,
                                                   repeats=train_rep, svgp=SVGP_, engine=eng)
            mu.append(m_); var.append(v_); aux.append(a_)
        mu, var, aux = torch.cat(mu), torch.cat(var), torch.cat(aux)
        mean_terms, var_terms = S.precompute_GP_params_SVGPVAE(mu, var, aux, SVGP_, engine=eng)
        K_mm, _, _ = eng.kernel_matrices(aux[:1])
        # :178 `tf.linalg.inv(K_mm)` WITHOUT jitter: with the linear kernels K_mm has rank <= L_action * L_character < m, and the
        # reference's LU with partial pivoting returns a (huge but finite) matrix where a Cholesky / no-pivot elimination -- the
        # library's SPD inverse -- has no answer: the library's row-pivoted LU inverse (lu.hip)
        K_mm_inv = S.general_inverse(K_mm)
        cg = []
        for lo in range(0, N_test - bt + 1, bt):
            _, _, loss = S.predict_SVGPVAE_sprites_test_character(
                (d_te[lo:lo + bt], a_te[lo:lo + bt]), VAE, SVGP_, repr_NN, mean_terms, var_terms, args.N_context, N_actions, bt,
                cgen_seg, cgen_rep, K_mm_inv, engine=eng)
            cg.append(float(loss))
        cgen = float(np.sum(cg) / (n_eval * (1 - args.N_context / N_actions)))
        log["cgen_mse"].append((epoch, cgen))
        print(f"Conditional generation MSE loss on test set for epoch {epoch}: {cgen}", flush=True)
        if chkpnt_dir:
            with open(chkpnt_dir + "pics/test_metrics.txt", "a") as f:
                f.write(f"{epoch + 1},{round(log['recon_mse_test'][-1][1], 4)},{round(cgen, 4)}\n")
            if args.save_model_weights:
                torch.save({"theta": eng.theta.cpu(), "adam_m": eng.adam_m.cpu(), "adam_v": eng.adam_v.cpu(),
                            "state": eng.state.cpu()}, chkpnt_dir + f"model_{epoch}.pt")
        ctx.barrier()
    log["total_time"] = time.time() - start
    log["world_size"], log["rank"] = ctx.world, ctx.rank
    log["rccl_ranks"] = comm.world_size if (comm is not None and comm_fallback is None) else 0
    say(f"Running time for {nr_epochs} epochs: {round(log['total_time'], 2)}")
    if root and getattr(args, "log_json", None):
        eng.stream.synchronize()
        json.dump(dict(log, theta=eng.theta.cpu().tolist(), param_order=list(eng.shapes)), open(args.log_json, "wt"))
    log["_engine"] = eng
    return log


_EVAL = {}


def _eval_engine(train_eng, N_actions):
    """Forward-only twin of the training engine for the test-character reconstruction (segment length = N_actions):
    shares the parameter vector, has its own workspace so the training state is untouched."""
    key = (id(train_eng), N_actions)
    if key not in _EVAL:
        e = S.SpritesStepEngine(S.spritesVAE(train_eng.L), S.sprites_representation_network(train_eng.Lc), train_eng.svgp,
                                b_max=train_eng.b_max, seg_len=N_actions, clip_qs=train_eng.clip_qs, geco=False,
                                params={k: v for k, v in train_eng.params.items()}, device=train_eng.dev)
        # share (not copy) the parameters: re-point the twin's views at the training engine's flat vector
        e.theta = train_eng.theta
        off = 0
        for k, shp in e.shapes.items():
            n = int(np.prod(shp))
            e.params[k] = e.theta[off:off + n].view(shp)
            off += n
        train_eng.svgp.inducing_index_points, train_eng.svgp.GPLVM_action, train_eng.svgp.se = \
            train_eng.params["inducing_index_points"], train_eng.params["GPLVM_action"], train_eng.params["se"]
        _EVAL[key] = e
    return _EVAL[key]


def main(argv=None):
    from .dp import DistContext
    args = build_parser().parse_args(argv)
    ctx = DistContext()
    try:
        return run_experiment_sprites_SVGPVAE(args, vars(args), ctx)
    finally:
        ctx.close()


if __name__ == "__main__":
    main()
