"""Moving-ball experiment on the HIP library.

Reference call surface mirrored here (eager float64 CUDA tensors instead of TF graph tensors):
  SVGP(titsias, num_inducing_points, fixed_inducing_points, tmin, tmax, vidlt, fixed_gp_params, name, jitter,
       ip_min, ip_max, GP_init)                                                   SVGPVAE_model.py:17-60
  build_SVGPVAE_elbo_graph(vid_batch, beta, svgp_x, svgp_y, clipping_qs=False)    SVGPVAE_model.py:638-715
  build_pearce_elbo_graphs(vid_batch, beta, type_elbo, lt, ..., GP_joint, GP_init) GPVAE_Pearce_model.py:89-236
  Make_path_batch / Make_Video_batch / build_video_batch_graph / MSE_rotation     utils.py:29-121,138-192,195-245
and `BallStepEngine.train_step` = the reference's `sess.run(optim_step)` (BALL_experiment.py:116-136, 213-217).

The two SVGP objects of the SVGPVAE ELBOs run on the shared sparse-GP stage kernels: the tmax frames of a video
are the rows, the videos of the batch are the channels (every video has the time stamps 1..tmax, so K_mm / K_nm
are shared), N_train = tmax, cfg.kl_form = 1 (the reference's ball KL, SVGPVAE_model.py:135-137), cfg.clip_pv = 2
(:693); one workspace per latent coordinate.  The MLPs are batched MFMA GEMMs + bias/tanh kernels, the Bernoulli
reconstruction term, the per-video ELBO assembly and TF1 Adam are kernels of ball.hip / optim.hip.  The reference
runs the ball experiment in float32; this build computes in float64 (a superset precision).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import STATE, STATE_LEN, MnistCfg, WsLayout, call

_F64 = torch.float64
OUT_ROWS = ("elbo", "recon", "KL_term", "inside_elbo", "ce_term", "inside_elbo_recon", "inside_elbo_kl")


# ---------------------------------------------------------------------------------------------------------
# host-side data utilities (numpy, as in the reference)
# ---------------------------------------------------------------------------------------------------------
def Make_path_batch(batch=40, tmax=30, lt=5, seed=None):
    """utils.py:29-56: x(t), y(t) ~ GP(0, SE(lt)); (batch, tmax, 2)."""
    T = np.arange(tmax)
    Sigma = np.exp(-0.5 / (lt * lt) * (T.reshape(-1, 1) - T.reshape(1, -1)) ** 2)
    np.random.seed(seed)
    traj = np.random.multivariate_normal(np.zeros(tmax), Sigma, (batch, 2))
    return np.transpose(traj, (0, 2, 1))


def _rasterize_np(traj_px, px, py, r):
    i = np.arange(px).reshape(1, 1, px, 1)
    j = np.arange(py).reshape(1, 1, 1, py)
    # utils.py:99-104: sq_x is laid along the LAST axis there (image[y][x]); kept as is
    sq = (j - traj_px[:, :, 0, None, None]) ** 2 + (i - traj_px[:, :, 1, None, None]) ** 2
    return 1 * (sq < r * r)


def Make_Video_batch(tmax=50, px=32, py=32, lt=5, batch=40, seed=1, r=3):
    """utils.py:59-121: (traj0 (batch,tmax,2), vid_batch (batch,tmax,px,py) of 0/1).  Like the reference, `seed` is
    not forwarded to Make_path_batch."""
    traj0 = Make_path_batch(batch=batch, tmax=tmax, lt=lt)
    traj = traj0.copy()
    traj[:, :, 0] = traj[:, :, 0] * (px / 5) + (0.5 * px)
    traj[:, :, 1] = traj[:, :, 1] * (py / 5) + (0.5 * py)
    return traj0, _rasterize_np(traj, px, py, r)


def MSE_rotation(X, Y, VX=None):
    """utils.py:195-245 (full_cholesky=False): affine least-squares map of latent paths X onto true paths Y."""
    batch, tmax, _ = X.shape
    Xa = np.hstack([X.reshape(batch * tmax, 2), np.ones((batch * tmax, 1))])
    W, MSE, _, _ = np.linalg.lstsq(Xa, Y.reshape(batch * tmax, 2), rcond=None)
    MSE = MSE[0] + MSE[1] if len(MSE) == 2 else np.nan
    X_rot = (Xa @ W).reshape(batch, tmax, 2)
    VX_rot = np.zeros((batch, tmax, 2, 2))
    if VX is not None:
        Wr = W[:2, :]
        VX_rot = np.einsum('ij,btj,kj->btik', Wr, VX, Wr)
    return X_rot, W, MSE, VX_rot


class VideoBatchSource:
    """build_video_batch_graph (utils.py:138-192): a fresh batch of ball videos per call, synthesised on the device:
    paths = chol(K_SE(lt) + 1e-5 I) N(0,1), scaled 0.2 px + 0.5 px, rasterised by svgp_ball_rasterize."""

    def __init__(self, tmax=50, px=32, py=32, lt=5, batch=1, seed=1, r=3, device="cuda:0"):
        assert px == py, "video batch graph assumes square frames"
        self.tmax, self.px, self.py, self.batch, self.r = tmax, px, py, batch, r
        self.dev = torch.device(device)
        t = torch.arange(tmax, dtype=_F64)
        K = torch.exp(-0.5 / (lt ** 2) * (t[:, None] - t[None, :]) ** 2) + 0.00001 * torch.eye(tmax, dtype=_F64)
        self.chol_K = torch.linalg.cholesky(K).to(self.dev)
        self.gen = torch.Generator(device=self.dev).manual_seed(seed)
        _lib.load_library()

    def __call__(self, stream=None):
        ran_Z = torch.randn(self.tmax, 2 * self.batch, dtype=_F64, device=self.dev, generator=self.gen)
        paths = (self.chol_K @ ran_Z).reshape(self.tmax, self.batch, 2).permute(1, 0, 2).contiguous()
        paths = paths * 0.2 * self.px + 0.5 * self.px
        vid = torch.empty(self.batch, self.tmax, self.px, self.py, dtype=_F64, device=self.dev)
        s = torch.cuda.current_stream(self.dev).cuda_stream if stream is None else stream
        call("svgp_ball_rasterize", self.batch * self.tmax, self.px, self.py, float(self.r), paths.data_ptr(),
             vid.data_ptr(), s)
        return vid


# ---------------------------------------------------------------------------------------------------------
# model objects
# ---------------------------------------------------------------------------------------------------------
class SVGP:
    """SVGPVAE_model.py:17-60.  Holds the initial inducing points / length scale; the engine owns the live values."""
    dtype = np.float64

    def __init__(self, titsias, num_inducing_points, fixed_inducing_points, tmin, tmax, vidlt, fixed_gp_params, name,
                 jitter, ip_min, ip_max, GP_init):
        self.titsias, self.num_inducing_points = bool(titsias), int(num_inducing_points)
        self.tmin, self.tmax, self.ip_min, self.ip_max, self.jitter = tmin, tmax, ip_min, ip_max, float(jitter)
        self.fixed_inducing_points, self.fixed_gp_params, self.name = bool(fixed_inducing_points), bool(fixed_gp_params), name
        lo, hi = (tmin, tmax) if fixed_inducing_points else (ip_min, ip_max)
        self.inducing_index_points = torch.linspace(float(lo), float(hi), self.num_inducing_points, dtype=_F64)
        self.l_GP = torch.tensor(float(vidlt if fixed_gp_params else GP_init), dtype=_F64)


def mlp_param_shapes(px=32, py=32, hidden=500):
    """Creation order of build_MLP_inference_graph / build_MLP_decoder_graph (VAE_utils.py:31-47, 79-91)."""
    P = px * py
    return [("encW1", (P, hidden)), ("encB1", (hidden,)), ("encW2", (hidden, 4)), ("encB2", (4,)),
            ("decW1", (2, hidden)), ("decB1", (hidden,)), ("decW2", (hidden, P)), ("decB2", (P,))]


def truncated_normal_mlp_params(px=32, py=32, hidden=500, seed=0):
    """tf.truncated_normal(stddev = 1/sqrt(fan_in)) weights (resampled beyond 2 sigma), zero biases."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shp in mlp_param_shapes(px, py, hidden):
        if len(shp) == 1:
            out[name] = np.zeros(shp)
            continue
        w = rng.standard_normal(shp)
        bad = np.abs(w) > 2
        while bad.any():
            w[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(w) > 2
        out[name] = w / math.sqrt(shp[0])
    return out


class BallStepEngine:
    """Buffers + kernel schedule of one moving-ball training step (SVGPVAE_Hensman / SVGPVAE_Titsias)."""

    def __init__(self, svgp_x, svgp_y, *, batch=35, tmax=30, px=32, py=32, hidden=500, clip_qs=False, beta=1.0,
                 lr=1e-3, clip_grad=False, device="cuda:0", params=None, seed=0):
        self.lib = _lib.load_library()
        if not torch.cuda.is_available():
            raise _lib.SvgpError("BallStepEngine needs a HIP device; there is no CPU execution path")
        if svgp_x.titsias != svgp_y.titsias or svgp_x.num_inducing_points != svgp_y.num_inducing_points:
            raise ValueError("svgp_x and svgp_y must agree on the ELBO branch and on the number of inducing points")
        self.dev = torch.device(device)
        self.B, self.T, self.px, self.py, self.P, self.H = batch, tmax, px, py, px * py, hidden
        self.m, self.titsias = svgp_x.num_inducing_points, svgp_x.titsias
        self.svgp = (svgp_x, svgp_y)
        self.clip_qs, self.clip_grad = bool(clip_qs), bool(clip_grad)
        self.stream = torch.cuda.Stream(device=self.dev)
        f64 = dict(dtype=_F64, device=self.dev)
        # ---- flat parameter vector: MLPs, then per coordinate inducing points and length scale
        self.shapes = dict(mlp_param_shapes(px, py, hidden))
        self.shapes.update(ip_x=(self.m,), l_x=(1,), ip_y=(self.m,), l_y=(1,))
        n_tot = sum(int(np.prod(s)) for s in self.shapes.values())
        self.theta, self.grad = torch.zeros(n_tot, **f64), torch.zeros(n_tot, **f64)
        self.adam_m, self.adam_v = torch.zeros(n_tot, **f64), torch.zeros(n_tot, **f64)
        self.params, self.grads, off = {}, {}, 0
        for k, s in self.shapes.items():
            n = int(np.prod(s))
            self.params[k], self.grads[k] = self.theta[off:off + n].view(s), self.grad[off:off + n].view(s)
            off += n
        init = dict(truncated_normal_mlp_params(px, py, hidden, seed) if params is None else params)
        for c, sv in zip("xy", self.svgp):
            init.setdefault(f"ip_{c}", sv.inducing_index_points)
            init.setdefault(f"l_{c}", sv.l_GP)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))     # zero-fills ran on torch's stream
        with torch.cuda.stream(self.stream):
            for k, s in self.shapes.items():
                v = init[k]
                self.params[k].copy_(torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v.detach().cpu(),
                                                     dtype=_F64).reshape(s))
        for c, sv in zip("xy", self.svgp):
            sv.inducing_index_points, sv.l_GP = self.params[f"ip_{c}"], self.params[f"l_{c}"]
        # ---- one GP workspace per latent coordinate: rows = frames, channels = videos
        self.cfg = MnistCfg(b=tmax, b_global=tmax, m=self.m, L=batch, M=1, n_obj=0, normalize_obj=0, clip_qs=0, geco=0,
                            train_ip=1, train_gp=1, train_ov=0, b_cap=tmax, clip_pv=2, n_pix=self.P,
                            titsias=int(self.titsias), kl_form=1, reserved_=0, N_train=float(tmax),
                            jitter=svgp_x.jitter, kappa_squared=0.0, alpha=0.0, rep_weight=1.0)
        if svgp_x.jitter != svgp_y.jitter:
            raise ValueError("svgp_x and svgp_y must use the same jitter")
        self.wl = WsLayout()
        call("svgp_mnist_ws_layout_get", C.byref(self.cfg), C.byref(self.wl))
        self.ws = [torch.zeros(self.wl.total, **f64) for _ in range(2)]
        self.state = torch.zeros(STATE_LEN, **f64)
        self.out = torch.zeros(len(OUT_ROWS), batch, **f64)
        self.times = torch.arange(1, tmax + 1, **f64)                   # SVGPVAE_model.py:663
        self.part = torch.zeros(int(self.lib.svgp_act_bwd_bias_scratch_elems(max(self.P, hidden))), **f64)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        st = torch.zeros(STATE_LEN, dtype=_F64)
        st[STATE["LR"]], st[STATE["BETA"]], st[STATE["LAGRANGE"]] = lr, beta, 1.0
        with torch.cuda.stream(self.stream):
            self.state.copy_(st)
        self.stream.synchronize()
        self.act = {}

    # ------------------------------------------------------------------ helpers
    def _v(self, c, name, shape):
        off = getattr(self.wl, name)
        return self.ws[c][off:off + int(np.prod(shape))].view(shape)

    def _gemm(self, ta, tb, M, N, K, A, lda, Bm, ldb, Cm, ldc):
        call("svgp_dgemm_batched", ta, tb, M, N, K, 1.0, A.data_ptr(), lda, 0, Bm.data_ptr(), ldb, 0, 0.0, Cm.data_ptr(),
             ldc, 0, 1, self.stream.cuda_stream)

    def scalars(self):
        self.stream.synchronize()
        st = self.state.cpu()
        return {k.lower(): float(st[i]) for k, i in STATE.items()}

    def set_scalars(self, **kw):
        self.stream.synchronize()
        st = self.state.cpu()
        for k, v in kw.items():
            st[STATE[k.upper()]] = float(v)
        with torch.cuda.stream(self.stream):
            self.state.copy_(st)
        self.stream.synchronize()

    # ------------------------------------------------------------------ one step
    def step(self, vid_batch, epsilon=None, adam=True, backward=True):
        """vid_batch (batch,tmax,px,py) float64 CUDA tensor; epsilon (batch,tmax,2) or None (on-device N(0,1)).
        Forward, reverse, optional gradient clip, TF1 Adam when `adam`, per-video ELBO terms and their means."""
        B, T, P, H, m = self.B, self.T, self.P, self.H, self.m
        assert tuple(vid_batch.shape) == (B, T, self.px, self.py)
        p, g, s = self.params, self.grads, self.stream.cuda_stream
        cp, st = C.byref(self.cfg), self.state.data_ptr()
        f64 = dict(dtype=_F64, device=self.dev)
        R = B * T
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.stream):
            X = vid_batch.to(self.dev, _F64).contiguous().view(R, P)
            # ---------------- encoder MLP (VAE_utils.py:26-55)
            h1 = torch.empty(R, H, **f64)
            self._gemm(0, 0, R, H, P, X, P, p["encW1"], H, h1, H)
            call("svgp_bias_act_fwd", R, H, 1, p["encB1"].data_ptr(), h1.data_ptr(), s)
            h2 = torch.empty(R, 4, **f64)
            self._gemm(0, 0, R, 4, H, h1, H, p["encW2"], 4, h2, 4)
            fld = lambda n: [self._v(c, n, (T, B)) for c in range(2)]
            mu, var_raw, var = fld("qnet_mu"), fld("qnet_var_raw"), fld("qnet_var")
            call("svgp_ball_head_fwd", B, T, int(self.clip_qs), p["encB2"].data_ptr(), h2.data_ptr(), mu[0].data_ptr(),
                 var_raw[0].data_ptr(), var[0].data_ptr(), mu[1].data_ptr(), var_raw[1].data_ptr(), var[1].data_ptr(), s)
            # ---------------- the two sparse GPs (SVGPVAE_model.py:673-681)
            eps_c = [None, None]
            if epsilon is not None:
                e = epsilon.to(self.dev, _F64)
                eps_c = [e[:, :, c].t().contiguous() for c in range(2)]
            for c, cn in enumerate("xy"):
                ws = self.ws[c].data_ptr()
                call("svgp_se1d_kernel_matrix_fwd", T, m, self.times.data_ptr(), p[f"ip_{cn}"].data_ptr(),
                     p[f"l_{cn}"].data_ptr(), self._v(c, "K", (1,)).data_ptr(), self._v(c, "Kn", (1,)).data_ptr(),
                     self._v(c, "knn", (1,)).data_ptr(), s)
                call("svgp_gp_stats_fwd", cp, ws, s)
                if self.titsias:
                    call("svgp_gp_titsias_stats", cp, ws, s)
                call("svgp_gp_factor_fwd", cp, ws, s)
                call("svgp_gp_posterior_fwd", cp, None if eps_c[c] is None else eps_c[c].data_ptr(), ws, st, s)
                if self.titsias:
                    call("svgp_gp_titsias_fwd", cp, ws, st, s)
                if c == 0:
                    call("svgp_state_add", st, STATE["RNG_CTR"], 1.0, s)      # fresh samples for the y coordinate
            # ---------------- decoder MLP + Bernoulli reconstruction term (:693-700)
            z = torch.empty(R, 2, **f64)
            call("svgp_ball_pack_z", B, T, self._v(0, "z", (1,)).data_ptr(), self._v(1, "z", (1,)).data_ptr(),
                 z.data_ptr(), s)
            g1 = torch.empty(R, H, **f64)
            self._gemm(0, 0, R, H, 2, z, 2, p["decW1"], H, g1, H)
            call("svgp_bias_act_fwd", R, H, 1, p["decB1"].data_ptr(), g1.data_ptr(), s)
            logits = torch.empty(R, P, **f64)
            self._gemm(0, 0, R, P, H, g1, H, p["decW2"], P, logits, P)
            call("svgp_bias_act_fwd", R, P, 0, p["decB2"].data_ptr(), logits.data_ptr(), s)
            pred, row_recon = torch.empty(R, P, **f64), torch.empty(R, **f64)
            dlog = torch.empty(R, P, **f64) if backward else None
            call("svgp_sigmoid_xent", R, P, 1.0 / B, logits.data_ptr(), X.data_ptr(), pred.data_ptr(),
                 row_recon.data_ptr(), None if dlog is None else dlog.data_ptr(), s)
            self.act = dict(pred=pred.view(B, T, self.px, self.py), z=z.view(B, T, 2))
            if backward:
                # ================ reverse: decoder
                self._gemm(1, 0, H, P, R, g1, H, dlog, P, g["decW2"], P)
                call("svgp_act_bwd_bias", R, P, 0, None, dlog.data_ptr(), self.part.data_ptr(), g["decB2"].data_ptr(), s)
                dg1 = torch.empty(R, H, **f64)
                self._gemm(0, 1, R, H, P, dlog, P, p["decW2"], P, dg1, H)
                call("svgp_act_bwd_bias", R, H, 1, g1.data_ptr(), dg1.data_ptr(), self.part.data_ptr(),
                     g["decB1"].data_ptr(), s)
                self._gemm(1, 0, 2, H, R, z, 2, dg1, H, g["decW1"], H)
                dz = torch.empty(R, 2, **f64)
                self._gemm(0, 1, R, 2, H, dg1, H, p["decW1"], H, dz, 2)
                call("svgp_ball_unpack_zbar", B, T, dz.data_ptr(), self._v(0, "zbar", (1,)).data_ptr(),
                     self._v(1, "zbar", (1,)).data_ptr(), s)
                # ================ reverse: the two GPs
                for c, cn in enumerate("xy"):
                    ws = self.ws[c].data_ptr()
                    call("svgp_gp_stats_bwd", cp, ws, st, s)
                    call("svgp_gp_factor_bwd", cp, ws, st, s)
                    call("svgp_gp_posterior_bwd", cp, ws, st, s)
                    if self.titsias:
                        call("svgp_gp_titsias_bwd", cp, ws, st, s)
                    call("svgp_se1d_kernel_matrix_bwd", T, m, self.times.data_ptr(), p[f"ip_{cn}"].data_ptr(),
                         p[f"l_{cn}"].data_ptr(), self._v(c, "Kbar", (1,)).data_ptr(), self._v(c, "Knbar", (1,)).data_ptr(),
                         g[f"ip_{cn}"].data_ptr(), g[f"l_{cn}"].data_ptr(), s)
                    if self.svgp[c].fixed_inducing_points:
                        g[f"ip_{cn}"].zero_()
                    if self.svgp[c].fixed_gp_params:
                        g[f"l_{cn}"].zero_()
                # ================ reverse: encoder
                dh2 = torch.empty(R, 4, **f64)
                yb, sb = fld("ybar"), fld("s2bar")
                call("svgp_ball_head_bwd", B, T, int(self.clip_qs), var_raw[0].data_ptr(), yb[0].data_ptr(),
                     sb[0].data_ptr(), var_raw[1].data_ptr(), yb[1].data_ptr(), sb[1].data_ptr(), dh2.data_ptr(), s)
                self._gemm(1, 0, H, 4, R, h1, H, dh2, 4, g["encW2"], 4)
                call("svgp_act_bwd_bias", R, 4, 0, None, dh2.data_ptr(), self.part.data_ptr(), g["encB2"].data_ptr(), s)
                dh1 = torch.empty(R, H, **f64)
                self._gemm(0, 1, R, H, 4, dh2, 4, p["encW2"], 4, dh1, H)
                call("svgp_act_bwd_bias", R, H, 1, h1.data_ptr(), dh1.data_ptr(), self.part.data_ptr(),
                     g["encB1"].data_ptr(), s)
                self._gemm(1, 0, P, H, R, X, P, dh1, H, g["encW1"], H)
                if self.clip_grad:                                      # BALL_experiment.py:125-127
                    call("svgp_clip_by_value", self.grad.numel(), 100000.0, self.grad.data_ptr(), s)
                if adam:
                    call("svgp_adam_tf1_step", self.theta.numel(), self.theta.data_ptr(), self.grad.data_ptr(),
                         self.adam_m.data_ptr(), self.adam_v.data_ptr(), st, 0.9, 0.999, 1e-8, s)
            call("svgp_ball_elbo_assemble", cp, self.ws[0].data_ptr(), self.ws[1].data_ptr(), row_recon.data_ptr(), st,
                 self.out.data_ptr(), s)
            call("svgp_ball_finalize", B, int(bool(adam and backward)), 1, self.out.data_ptr(), st, s)
        return self

    train_step = step

    def outputs(self):
        """build_SVGPVAE_elbo_graph's return tuple for the last step (the trailing globals() slot holds the engine)."""
        self.stream.synchronize()
        B, T, m = self.B, self.T, self.m
        o = {k: self.out[i].clone() for i, k in enumerate(OUT_ROWS)}
        st2 = lambda n: torch.stack([self._v(c, n, (T, B)).t() for c in range(2)], 2).contiguous()
        full_p_mu, full_p_var, qnet_mu, qnet_var = st2("p_m"), st2("p_v"), st2("qnet_mu"), st2("qnet_var")
        p = self.params
        # mean over videos of the posterior covariance at the frame times (diagnostic of :683-685), computed from the
        # workspace matrices B_b = K_nn - K_nm K_mm^-1 K_mn + K_nm Sigma_b^-1 K_mn; reporting only, O(batch tmax^2 m)
        cov = []
        for c, cn in enumerate("xy"):
            Kn, Ki, Si = self._v(c, "Kn", (T, m)), self._v(c, "Ki", (m, m)), self._v(c, "Si", (B, m, m))
            t = self.times
            Knn = torch.exp(-0.5 * (t[:, None] - t[None, :]) ** 2 / p[f"l_{cn}"] ** 2)
            cov.append(Knn - Kn @ Ki @ Kn.t() + Kn @ Si.mean(0) @ Kn.t())
        return (o["elbo"], o["recon"], o["KL_term"], o["inside_elbo"], o["ce_term"], full_p_mu, full_p_var, qnet_mu,
                qnet_var, self.act["pred"], p["l_x"][0].clone(), p["l_y"][0].clone(), o["inside_elbo_recon"],
                o["inside_elbo_kl"], p["ip_x"].clone(), p["ip_y"].clone(), cov[0], cov[1], self)


def build_SVGPVAE_elbo_graph(vid_batch, beta, svgp_x, svgp_y, clipping_qs=False, epsilon=None, params=None, engine=None):
    """SVGPVAE_model.py:638-715: one forward pass on the HIP library; returns the reference's 19-tuple (last slot =
    the engine instead of globals()).  `epsilon` (batch,tmax,2) and `params` (MLP weights) are injectable for parity."""
    B, T, px, py = vid_batch.shape
    eng = engine or getattr(svgp_x, "_engine", None)
    if eng is None or (eng.B, eng.T, eng.px, eng.py) != (B, T, px, py) or eng.clip_qs != bool(clipping_qs):
        hidden = 500 if params is None else int(np.asarray(params["encB1"]).size)
        eng = BallStepEngine(svgp_x, svgp_y, batch=B, tmax=T, px=px, py=py, hidden=hidden, clip_qs=clipping_qs,
                             beta=float(beta), params=params)
        svgp_x._engine = eng
    eng.set_scalars(beta=float(beta))
    eng.step(vid_batch, epsilon, adam=False, backward=False)
    return eng.outputs()
