"""The reference's `SVGPVAE_model.py` call surface for the rotated-MNIST Hensman path, executed by
libsvgpvae_hip.so on MI355X.  Eager float64 CUDA tensors replace TF graph tensors; the N(0,1)
draw `epsilon` and all weights are injectable so results can be compared on identical inputs.

Mirrored names (reference file:line):
  mnistSVGP(titsias, fixed_inducing_points, initial_inducing_points, fixed_gp_params,
            object_vectors_init, name, jitter, N_train, L, K_obj_normalize)      SVGPVAE_model.py:383-384
    .kernel_matrix(x, y, x_inducing=True, y_inducing=True, diag_only=False)      :427
    .approximate_posterior_params(index_points_test, index_points_train, y, noise) -> (mean, B, mu_hat, A_hat)  :303
    .variational_loss(x, y, mu_hat, A_hat, noise) -> (L3 sum term, KL term)       :220
    .variable_summary()                                                           :478
  forward_pass_SVGPVAE(data_batch, beta, vae, svgp, C_ma, lagrange_mult, alpha, kappa,
                       clipping_qs=False, GECO=False, ...) -> 16-tuple            :823-936
  batching_encode_SVGPVAE(data_batch, vae, clipping_qs=False, ...)                :939-968
Additional (no reference counterpart - TF's tf.gradients + AdamOptimizer live in the driver):
  gradients_SVGPVAE(...) and train_step_SVGPVAE(...).

  bacthing_predict_SVGPVAE_rotated_mnist(test_data_batch, vae, svgp, qnet_mu, qnet_var, aux_data_train)  :1026-1083
    .mean_vector_bias_analysis(index_points, y, noise)                            :345-378
titsias=True selects the SVGPVAE_Titsias inside-ELBO (:246-259), computed in m x m space (Woodbury; see
gp_titsias.hip).  kernel_matrix accepts every (x_inducing, y_inducing, diag_only) pattern on arbitrary row sets;
approximate_posterior_params accepts test points != train points.

The SPRITES and moving-ball names of the reference's module resolve here as well (SPRITES_experiment.py:14-16,
BALL_experiment.py:14): spritesSVGP (:487), precompute_GP_params_SVGPVAE (:989), aux_data_SVGPVAE_sprites (:1086),
predict_SVGPVAE_sprites_test_character (:1118), SVGP (:17), build_SVGPVAE_elbo_graph (:638).  Their code lives beside the
engines that run them (sprites.py, ball.py); forward_pass_SVGPVAE / batching_encode_SVGPVAE dispatch on `repr_NN`.
"""
import math

import numpy as np
import torch

from .engine import MnistStepEngine

_F64 = torch.float64


class mnistSVGP:
    def __init__(self, titsias, fixed_inducing_points, initial_inducing_points, fixed_gp_params,
                 object_vectors_init, name, jitter, N_train, L, K_obj_normalize=False, device="cuda:0"):
        self.dtype = _F64
        self.titsias = bool(titsias)
        self.jitter = float(jitter)
        self.N_train = float(N_train)
        self.L = L
        self.K_obj_normalize = bool(K_obj_normalize)
        self.fixed_inducing_points = bool(fixed_inducing_points)
        self.fixed_gp_params = bool(fixed_gp_params)
        self.name = name
        self.device = device
        ip = torch.as_tensor(np.asarray(initial_inducing_points), dtype=_F64)
        self.nr_inducing = ip.shape[0]
        self.inducing_index_points = ip.clone()
        self.l_GP = torch.tensor(1.0, dtype=_F64)          # SVGPVAE_model.py:409-413
        self.amplitude = torch.tensor(1.0, dtype=_F64)
        self.object_vectors = None if object_vectors_init is None else \
            torch.as_tensor(np.asarray(object_vectors_init), dtype=_F64).clone()
        self._rt = None

    # -- parameters as a dict in the engine's naming
    def _params(self):
        p = {"inducing_index_points": self.inducing_index_points, "l_GP": self.l_GP, "amplitude": self.amplitude}
        if self.object_vectors is not None:
            p["object_vectors"] = self.object_vectors
        return p

    def _gp_engine(self, b, L):
        """A private engine for the stand-alone GP methods (per-channel API, L = 1)."""
        M = self.inducing_index_points.shape[1] - 2
        n_obj = 0 if self.object_vectors is None else self.object_vectors.shape[0]
        eng = MnistStepEngine(self.nr_inducing, L, M, n_obj, N_train=self.N_train, jitter=self.jitter,
                              clip_qs=False, geco=False, K_obj_normalize=self.K_obj_normalize, titsias=self.titsias,
                              b_max=b, device=self.device)
        eng.load_params({k: v for k, v in self._params().items()})
        return eng

    def kernel_matrix(self, x, y, x_inducing=True, y_inducing=True, diag_only=False):
        """K(x, y), SVGPVAE_model.py:427-476, for any argument pattern: rows (n, 2+M) [id, angle, o_1..o_M]; a side that
        is not `*_inducing` takes its object vector from the GPLVM table by the row's id when there is a table
        (:451,455), otherwise columns 2: are used (:444-445).  diag_only -> vector k(x_i, y_i) (:458-467)."""
        import ctypes as C
        from ._lib import call, load_library
        load_library()
        if not torch.cuda.is_available():
            from ._lib import SvgpError
            raise SvgpError("kernel_matrix needs a HIP device; there is no CPU execution path")
        dev = torch.device(self.device)
        M = self.inducing_index_points.shape[1] - 2
        dx = x.to(dev, _F64).contiguous()
        dy = dx if y is x else y.to(dev, _F64).contiguous()
        if dx.shape[1] != 2 + M or dy.shape[1] != 2 + M:
            raise ValueError(f"rows must have {2 + M} columns [id, angle, object vector]")
        nx, ny = dx.shape[0], dy.shape[0]
        if diag_only and nx != ny:
            raise ValueError("diag_only needs the same number of rows in x and y")
        table = None if self.object_vectors is None else self.object_vectors.to(dev, _F64).contiguous()
        xg = int(table is not None and not x_inducing)
        yg = int(table is not None and not y_inducing)
        ls = self.l_GP.to(dev, _F64).reshape(1).contiguous()
        amp = self.amplitude.to(dev, _F64).reshape(1).contiguous()
        out = torch.empty((nx,) if diag_only else (nx, ny), dtype=_F64, device=dev)
        call("svgp_kernel_matrix_xy", M, int(self.K_obj_normalize), nx, dx.data_ptr(), xg, ny, dy.data_ptr(), yg,
             None if table is None else table.data_ptr(), ls.data_ptr(), amp.data_ptr(), int(diag_only),
             out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        return out

    def _channel(self, index_points_train, y, noise, index_points_test=None):
        """Runs the GP stages for one channel.  Statistics S, v come from the train rows; the factor stage and the
        per-row stage run at the test rows (= the train rows when index_points_test is None) with c = N_train / (number
        of train rows), the reference's `self.N_train / b` (:316,328).  The workspace layout depends only on the row
        capacity, so both passes share offsets."""
        import ctypes as C
        from ._lib import call
        b = index_points_train.shape[0]
        bt = b if index_points_test is None else index_points_test.shape[0]
        eng = self._gp_engine(b, 1)
        dev = eng.device
        d_aux = index_points_train.to(dev, _F64).contiguous()
        d_aux_te = None if index_points_test is None else index_points_test.to(dev, _F64).contiguous()
        eng.set_batch_size(b, b)
        eng.ws_view("qnet_mu", (b, 1)).copy_(y.to(dev, _F64).reshape(b, 1))
        eng.ws_view("qnet_var", (b, 1)).copy_(noise.to(dev, _F64).reshape(b, 1))
        zeros = torch.zeros(b, 1, dtype=_F64, device=dev)
        th, ws, st, s = eng.theta.data_ptr(), eng.ws.data_ptr(), eng.state.data_ptr(), eng.stream.cuda_stream
        with torch.cuda.stream(eng.stream):
            eng.stream.wait_stream(torch.cuda.current_stream(dev))
            cfg = C.byref(eng.cfg)
            call("svgp_kernel_matrix_fwd", cfg, th, d_aux.data_ptr(), ws, s)
            call("svgp_gp_stats_fwd", cfg, ws, s)
            if self.titsias:
                call("svgp_gp_titsias_stats", cfg, ws, s)
            if d_aux_te is None:
                call("svgp_gp_factor_fwd", cfg, ws, s)
                call("svgp_gp_posterior_fwd", cfg, zeros.data_ptr(), ws, st, s)
                if self.titsias:
                    call("svgp_gp_titsias_fwd", cfg, ws, st, s)
            else:
                # per-row quantities of the factor / posterior stages at the test rows, in chunks of at most b rows
                # (a rank's rows never exceed the global rows); the L3 / CE integrands those kernels also write use
                # y / noise of the leading train rows and are not part of this method's result
                p_m = torch.empty(bt, dtype=_F64, device=dev)
                p_v = torch.empty(bt, dtype=_F64, device=dev)
                for lo in range(0, bt, b):
                    n = min(b, bt - lo)
                    eng.set_batch_size(n, b)
                    cfg = C.byref(eng.cfg)
                    call("svgp_kernel_matrix_fwd", cfg, th, d_aux_te[lo:lo + n].data_ptr(), ws, s)
                    call("svgp_gp_factor_fwd", cfg, ws, s)
                    call("svgp_gp_posterior_fwd", cfg, zeros.data_ptr(), ws, st, s)
                    p_m[lo:lo + n].copy_(eng.ws_view("p_m", (n,)))
                    p_v[lo:lo + n].copy_(eng.ws_view("p_v", (n,)))
        eng.synchronize()
        if d_aux_te is not None:
            return eng, (p_m, p_v)
        return eng, (eng.ws_view("p_m", (b,)).clone(), eng.ws_view("p_v", (b,)).clone())

    def approximate_posterior_params(self, index_points_test, index_points_train=None, y=None, noise=None):
        """(mean_vector, B, mu_hat, A_hat) of q_S for one latent channel (SVGPVAE_model.py:303-343): posterior mean and
        (diagonal) covariance at index_points_test given (index_points_train, y, noise)."""
        if index_points_train is None:
            index_points_train = index_points_test
        same = index_points_test is index_points_train or (index_points_test.shape == index_points_train.shape
                                                           and torch.equal(index_points_test, index_points_train))
        eng, (p_m, p_v) = self._channel(index_points_train, y, noise, None if same else index_points_test)
        m = self.nr_inducing
        return p_m, p_v, eng.ws_view("mu_hat", (m,)).clone(), eng.ws_view("A", (m, m)).clone()

    def mean_vector_bias_analysis(self, index_points, y=None, noise=None):
        """SVGPVAE_model.py:345-378 (Supplementary C.4): c K_mm Sigma_l^-1 K_mb (y / noise), c = N_train / b -- the same
        expression as mu_hat of approximate_posterior_params (:339-340)."""
        eng, _ = self._channel(index_points, y, noise)
        return eng.ws_view("mu_hat", (self.nr_inducing,)).clone()

    def variational_loss(self, x, y, mu_hat, A_hat, noise=None):
        """(L_3 sum term, KL term) of one channel (SVGPVAE_model.py:261-301).  mu_hat / A_hat must be the
        ones approximate_posterior_params returns for the same (x, y, noise) - the only use in the
        reference (:869-873); they are recomputed on device."""
        eng, _ = self._channel(x, y, noise)
        b = x.shape[0]
        if self.titsias:   # (L_2, 0)  :246-259
            sc = eng.ws_view("tit_scal", (3,))       # [log det Sigma2, v2.t2, row sum]  (L = 1)
            l2 = -0.5 * (b * 1.8378770664093453 + sc[2] + sc[0] - eng.ws_view("ldK", (1,))[0] - sc[1])
            return l2.clone(), torch.zeros((), dtype=_F64, device=eng.device)
        p = 1.0 / eng.ws_view("qnet_var", (b,))
        l3 = -0.5 * (torch.sum(p * eng.ws_view("d", (b,))) + torch.sum(torch.log(eng.ws_view("qnet_var", (b,))))
                     + b * 1.8378770664093453)
        return l3, eng.ws_view("KL", (1,))[0].clone()

    def variable_summary(self):
        """SVGPVAE_model.py:478-485: (l_GP, amplitude, object_vectors, inducing_index_points)."""
        return self.l_GP, self.amplitude, self.object_vectors, self.inducing_index_points


# ---------------------------------------------------------------------------------------------
class _Runtime:
    """Binds a (vae, svgp) pair to one MnistStepEngine: copies their parameters into the flat vector
    and re-points the objects' attributes at views of it, so optimiser updates are visible to both."""

    def __init__(self, vae, svgp, clipping_qs, GECO, kappa, alpha_flag=0.99, b_max=256, lr=1e-3, beta=0.001,
                 rank=0, world_size=1, split_grad_exchange=False):
        M = svgp.inducing_index_points.shape[1] - 2
        n_obj = 0 if svgp.object_vectors is None else svgp.object_vectors.shape[0]
        self.key = (bool(clipping_qs), bool(GECO), float(kappa))
        self.eng = MnistStepEngine(svgp.nr_inducing, vae.L, M, n_obj, N_train=svgp.N_train, jitter=svgp.jitter,
                                   clip_qs=clipping_qs, geco=GECO, K_obj_normalize=svgp.K_obj_normalize,
                                   titsias=svgp.titsias,
                                   kappa_squared=float(kappa) ** 2, alpha=alpha_flag, beta=beta, lr=lr,
                                   train_ip=not svgp.fixed_inducing_points, train_gp=not svgp.fixed_gp_params,
                                   train_ov=svgp.object_vectors is not None, b_max=b_max, device=svgp.device,
                                   rank=rank, world_size=world_size, split_grad_exchange=split_grad_exchange)
        params = dict(vae.params)
        params.update(svgp._params())
        self.eng.load_params(params)
        vae.params = {k: self.eng.params[k] for k in vae.params}
        vae._engine = self.eng
        svgp.inducing_index_points = self.eng.params["inducing_index_points"]
        svgp.l_GP = self.eng.params["l_GP"]
        svgp.amplitude = self.eng.params["amplitude"]
        if svgp.object_vectors is not None:
            svgp.object_vectors = self.eng.params["object_vectors"]
        svgp._rt = self


def _runtime(vae, svgp, clipping_qs, GECO, kappa, b, **kw):
    """The engine bound to (vae, svgp).  It is rebuilt when the flags change or more rows are needed than it was sized
    for (conditional generation over the whole train set after training at b = 256); a rebuild carries the COMPLETE
    training state over: parameters, Adam moments, and the device state vector (global step, GECO C_ma / lagrange
    multiplier / first-step alpha, learning rate, beta, RNG counter) -- replacing the engine must not restart the
    optimiser or the GECO trajectory."""
    rt = svgp._rt
    if rt is None or rt.key != (bool(clipping_qs), bool(GECO), float(kappa)) or rt.eng.b_max < b:
        old = rt
        if old is not None:
            old.eng.synchronize()
            vae.params = {k: v.detach().cpu().clone() for k, v in vae.params.items()}
            vae._engine = None
            for k in ("inducing_index_points", "l_GP", "amplitude", "object_vectors"):
                v = getattr(svgp, k)
                if v is not None:
                    setattr(svgp, k, v.detach().cpu().clone())
            kw.setdefault("alpha_flag", old.eng.base["alpha"])
            kw.setdefault("rank", old.eng.rank)
            kw.setdefault("world_size", old.eng.world_size)
            kw.setdefault("split_grad_exchange", bool(old.eng.base.get("split_grad_exchange", 0)))
        rt = _Runtime(vae, svgp, clipping_qs, GECO, kappa, b_max=max(b, 256 if old is None else old.eng.b_max), **kw)
        if old is not None:
            new, prev = rt.eng, old.eng
            with torch.cuda.stream(new.stream):
                new.adam_m.copy_(prev.adam_m)
                new.adam_v.copy_(prev.adam_v)
                new.state.copy_(prev.state)
            new.synchronize()
            if rt.key[1] != old.key[1]:       # GECO switched on / off: alpha-for-next-step follows the new objective
                new.set_scalars(alpha=0.0 if (rt.key[1] and new.scalars()["adam_t"] == 0) else new.base["alpha"])
    return rt


def _prepare(data_batch, beta, vae, svgp, C_ma, lagrange_mult, alpha, kappa, clipping_qs, GECO, epsilon,
             b_global=None, **kw):
    images, aux_data = data_batch
    b = images.shape[0]
    rt = _runtime(vae, svgp, clipping_qs, GECO, kappa, b, **kw)
    eng = rt.eng
    eng.set_batch_size(b, b_global)
    eng.set_scalars(beta=float(beta), c_ma=float(C_ma), lagrange=float(lagrange_mult), alpha=float(alpha))
    dev = eng.device
    eng.bind(images.to(dev, _F64).contiguous(), aux_data.to(dev, _F64).contiguous(),
             None if epsilon is None else epsilon.to(dev, _F64).contiguous())
    return eng, b


def _tuple16(eng, b):
    L = eng.base["L"]
    sc = eng.scalars()
    t = lambda v: torch.tensor(v, dtype=_F64, device=eng.device)
    w = lambda n, s: eng.ws_view(n, s).clone()
    return (t(sc["elbo"]), t(sc["recon_loss"]), t(sc["kl_term"]), t(sc["inside_elbo"]), t(sc["ce_term"]),
            w("p_m", (b, L)), w("p_v", (b, L)), w("qnet_mu", (b, L)), w("qnet_var", (b, L)),
            w("recon", (b, 28, 28, 1)), t(sc["inside_recon"]), t(sc["inside_kl"]), w("z", (b, L)),
            t(sc["c_ma"]), t(sc["lagrange"]), t(1.0))


def forward_pass_SVGPVAE(data_batch, beta, vae, svgp, C_ma, lagrange_mult, alpha, kappa,
                         clipping_qs=False, GECO=False, repr_NN=None, segment_ids=None, repeats=None,
                         bias_analysis=False, epsilon=None):
    """SVGPVAE_model.py:823-936.  Returns (elbo, recon_loss, KL_term, inside_elbo, ce_term, p_m, p_v, qnet_mu,
    qnet_var, recon_images, inside_elbo_recon, inside_elbo_kl, latent_samples, C_ma, lagrange_mult,
    mean_vectors).  With GECO the `elbo` slot holds the GECO loss and recon_loss is kappa^2-shifted, as in
    the reference (:909-913).  epsilon (b,L): the N(0,1) draw of :901; None -> drawn on device."""
    if repr_NN is not None:          # :861-863: aux_data computed per batch by the representation network (SPRITES)
        from . import sprites
        return sprites.forward_pass_SVGPVAE(data_batch, beta, vae, svgp, C_ma, lagrange_mult, alpha, kappa,
                                            clipping_qs=clipping_qs, GECO=GECO, repr_NN=repr_NN, segment_ids=segment_ids,
                                            repeats=repeats, bias_analysis=bias_analysis, epsilon=epsilon)
    eng, b = _prepare(data_batch, beta, vae, svgp, C_ma, lagrange_mult, alpha, kappa, clipping_qs, GECO, epsilon)
    with torch.cuda.stream(eng.stream):
        eng.phase(0)
        eng.phase(1)
        eng.phase(2)                 # gradients come with the forward at negligible cost
        eng.phase(3, adam=False)
    eng.synchronize()
    out = _tuple16(eng, b)
    if bias_analysis:
        # :927-931: one mean_vector_bias_analysis per channel = c K_mm Sigma_l^-1 K_mb (y_l / noise_l) = mu_hat_l, which
        # the factor stage of the step has just computed for every channel
        mu_hat = eng.ws_view("mu_hat", (eng.base["L"], eng.base["m"])).clone()
        out = out[:15] + ([mu_hat[l] for l in range(eng.base["L"])],)
    return out


def gradients_SVGPVAE(vae, svgp):
    """d(minimised objective)/d(parameters) of the last forward_pass_SVGPVAE / train_step_SVGPVAE
    (tf.gradients of MNIST_experiment.py:202-205: GECO -> `elbo` slot, else -elbo)."""
    return {k: v.clone() for k, v in svgp._rt.eng.grads().items()}


def train_step_SVGPVAE(data_batch, beta, vae, svgp, alpha, kappa, lr, clipping_qs=False, GECO=False,
                       epsilon=None, reset=False):
    """One optimiser step = the reference's `sess.run([optim_step, ...])` (MNIST_experiment.py:334-340):
    forward, reverse, TF1 Adam, and the GECO state carry (C_ma, lagrange_mult, first-step alpha=0) kept on
    device between calls.  Returns the 16-tuple of the step."""
    images = data_batch[0]
    rt = _runtime(vae, svgp, clipping_qs, GECO, kappa, images.shape[0], alpha_flag=alpha)
    eng = rt.eng
    eng.base["alpha"] = float(alpha)      # cfg.alpha = the moving-average constant from the 2nd step on; not in rt.key
    if reset:
        eng.reset_state()
    eng.set_batch_size(images.shape[0])
    eng.set_scalars(beta=float(beta), lr=float(lr))
    dev = eng.device
    eng.bind(images.to(dev, _F64).contiguous(), data_batch[1].to(dev, _F64).contiguous(),
             None if epsilon is None else epsilon.to(dev, _F64).contiguous())
    eng.run(adam=True)
    eng.synchronize()
    return _tuple16(eng, images.shape[0])


def batching_encode_SVGPVAE(data_batch, vae, clipping_qs=False, repr_nn=None, segment_ids=None, repeats=None):
    """SVGPVAE_model.py:939-968: (qnet_mu, qnet_var, aux_data)."""
    if repr_nn is not None:          # :953-956 (SPRITES form; runs on the engine attached to `repr_nn` / `vae`)
        from . import sprites
        return sprites.batching_encode_SVGPVAE(data_batch, vae, clipping_qs, repr_nn, segment_ids, repeats)
    images, aux_data = data_batch
    mu, var = vae.encode(images)
    if clipping_qs:
        var = torch.clamp(var, 1e-3, 10.0)
    return mu, var, aux_data


def bacthing_predict_SVGPVAE_rotated_mnist(test_data_batch, vae, svgp, qnet_mu, qnet_var, aux_data_train,
                                           epsilon=None):
    """SVGPVAE_model.py:1026-1083 (sic spelling): conditional generation for a test batch given the
    encodings (qnet_mu, qnet_var) (N,L) and aux data (N,2+M) of the whole train set.  Returns
    (recon_images_test, recon_loss) with recon_loss = sum (x - x_hat)^2 / (w*h)  (:1076-1080).

    Device schedule: kernel matrices + statistics + factor stage over the N train rows (b = N, so
    N_train/b is the reference's `self.N_train / b`, :316,328), then kernel matrices, per-sample stage and
    decoder over the test rows with the same channel factors (the capacity-based workspace layout keeps
    S, Sigma_l^-1, t at fixed offsets).  epsilon (b_test, L): the N(0,1) draw of :1057; None -> on device."""
    import ctypes as C
    from ._lib import call
    images_test, aux_test = test_data_batch
    N, bt, L = aux_data_train.shape[0], aux_test.shape[0], vae.L
    rt = svgp._rt
    if rt is None or rt.eng.b_max < max(N, bt):
        # a larger engine; _runtime carries parameters, Adam moments and the GECO / step state of the old one over
        rt = _runtime(vae, svgp, True if rt is None else rt.key[0], False if rt is None else rt.key[1],
                      math.sqrt(0.020) if rt is None else rt.key[2], max(N, bt))
    eng = rt.eng
    dev = eng.device
    saved = (eng.cfg.b, eng.cfg.b_global)
    th, ws, st, s = eng.theta.data_ptr(), eng.ws.data_ptr(), eng.state.data_ptr(), eng.stream.cuda_stream
    d_aux_tr = aux_data_train.to(dev, _F64).contiguous()
    d_aux_te = aux_test.to(dev, _F64).contiguous()
    d_img_te = images_test.to(dev, _F64).contiguous()
    d_eps = None if epsilon is None else epsilon.to(dev, _F64).contiguous()
    # ---- train rows: statistics and channel factors
    eng.set_batch_size(N, N)
    eng.ws_view("qnet_mu", (N, L)).copy_(qnet_mu.to(dev, _F64))
    eng.ws_view("qnet_var", (N, L)).copy_(qnet_var.to(dev, _F64))
    with torch.cuda.stream(eng.stream):
        eng.stream.wait_stream(torch.cuda.current_stream(dev))
        cfg = C.byref(eng.cfg)
        call("svgp_kernel_matrix_fwd", cfg, th, d_aux_tr.data_ptr(), ws, s)
        call("svgp_gp_stats_fwd", cfg, ws, s)
        # ---- test rows: same global row count (c = N_train / N), different local rows
        eng.set_batch_size(bt, N)
        cfg = C.byref(eng.cfg)
        call("svgp_kernel_matrix_fwd", cfg, th, d_aux_te.data_ptr(), ws, s)
        call("svgp_gp_factor_fwd", cfg, ws, s)            # channel factors + q for the test rows
        call("svgp_gp_posterior_fwd", cfg, None if d_eps is None else d_eps.data_ptr(), ws, st, s)
        call("svgp_mnist_decoder_fwd", cfg, th, d_img_te.data_ptr(), ws, s)
    eng.synchronize()
    recon = eng.ws_view("recon", (bt, 28, 28, 1)).clone()
    eng.set_batch_size(*saved)
    recon_loss = torch.sum((d_img_te - recon) ** 2) / 784.0
    return recon, recon_loss


# the reference's SVGPVAE_model.py also defines the SPRITES and moving-ball objects (see the module docstring)
from .sprites import (spritesSVGP, precompute_GP_params_SVGPVAE, aux_data_SVGPVAE_sprites,  # noqa: E402,F401
                      predict_SVGPVAE_sprites_test_character)
from .ball import SVGP, build_SVGPVAE_elbo_graph  # noqa: E402,F401
