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


def post_process_full_cholesky(arr, tmax):
    """utils.py:248-259: (batch, tmax, 2 tmax) lower-triangular factors [L_x | L_y] -> (batch, tmax, 2) diagonals of L L^T."""
    Lx, Ly = np.tril(arr[:, :, :tmax]), np.tril(arr[:, :, tmax:])
    return np.stack([(Lx * Lx).sum(2), (Ly * Ly).sum(2)], axis=2)


def MSE_rotation(X, Y, VX=None, full_cholesky=False):
    """utils.py:195-245: affine least-squares map of latent paths X onto true paths Y; VX = per-point variances (batch, tmax, 2),
    or with `full_cholesky` the (batch, tmax, 2 tmax) Cholesky factors they are the diagonals of."""
    batch, tmax, _ = X.shape
    Xa = np.hstack([X.reshape(batch * tmax, 2), np.ones((batch * tmax, 1))])
    W, MSE, _, _ = np.linalg.lstsq(Xa, Y.reshape(batch * tmax, 2), rcond=None)
    MSE = MSE[0] + MSE[1] if len(MSE) == 2 else np.nan
    X_rot = (Xa @ W).reshape(batch, tmax, 2)
    VX_rot = np.zeros((batch, tmax, 2, 2))
    if VX is not None:
        if full_cholesky:
            VX = post_process_full_cholesky(VX, tmax)
        Wr = W[:2, :]
        VX_rot = np.einsum('ij,btj,kj->btik', Wr, VX, Wr)
    return X_rot, W, MSE, VX_rot


def _mm(A, B, tb=False, stream=None):
    """A (M,K) @ B (K,N) (or B^T with tb: B stored (N,K)) on the library's float64 MFMA GEMM (svgp_dgemm_batched)."""
    A, B = A.contiguous(), B.contiguous()
    M, K = A.shape
    N = B.shape[0] if tb else B.shape[1]
    out = torch.empty(M, N, dtype=_F64, device=A.device)
    s = torch.cuda.current_stream(A.device).cuda_stream if stream is None else stream
    call("svgp_dgemm_batched", 0, int(tb), M, N, K, 1.0, A.data_ptr(), A.shape[1], 0, B.data_ptr(), B.shape[1], 0, 0.0,
         out.data_ptr(), N, 0, 1, s)
    return out


class VideoBatchSource:
    """build_video_batch_graph (utils.py:138-192): a fresh batch of ball videos per call, synthesised on the device:
    paths = chol(K_SE(lt) + 1e-5 I) N(0,1), scaled 0.2 px + 0.5 px, rasterised by svgp_ball_rasterize."""

    def __init__(self, tmax=50, px=32, py=32, lt=5, batch=1, seed=1, r=3, device="cuda:0"):
        assert px == py, "video batch graph assumes square frames"
        self.tmax, self.px, self.py, self.batch, self.r = tmax, px, py, batch, r
        self.dev = torch.device(device)
        t = torch.arange(tmax, dtype=_F64)
        K = torch.exp(-0.5 / (lt ** 2) * (t[:, None] - t[None, :]) ** 2) + 0.00001 * torch.eye(tmax, dtype=_F64)
        # tf.linalg.cholesky (utils.py:170) on the library's own factorisation (svgp_potrf_batched: lower factor in place)
        lib = _lib.load_library()
        Kd = K.to(self.dev).contiguous()
        ld = torch.empty(1, dtype=_F64, device=self.dev)
        work = torch.empty(max(int(lib.svgp_potrf_workspace_elems(tmax, 1)), 1), dtype=_F64, device=self.dev)
        call("svgp_potrf_batched", tmax, 1, Kd.data_ptr(), tmax, tmax * tmax, ld.data_ptr(), work.data_ptr(),
             torch.cuda.current_stream(self.dev).cuda_stream)
        torch.cuda.current_stream(self.dev).synchronize()
        self.chol_K = Kd
        self.gen = torch.Generator(device=self.dev).manual_seed(seed)

    def __call__(self, stream=None):
        ran_Z = torch.randn(self.tmax, 2 * self.batch, dtype=_F64, device=self.dev, generator=self.gen)
        paths = _mm(self.chol_K, ran_Z).reshape(self.tmax, self.batch, 2).permute(1, 0, 2).contiguous()
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


class _BallMlpEngine:
    """Shared part of the two moving-ball engines: flat parameters, the MLP encoder / decoder
    (build_MLP_inference_graph / build_MLP_decoder_graph, VAE_utils.py:9-96) forward and reverse, the Bernoulli
    reconstruction term, gradient clip + TF1 Adam (BALL_experiment.py:116-136)."""

    def _init_common(self, extra_shapes, init, *, batch, tmax, px, py, hidden, beta, lr, clip_grad, device, seed):
        self.lib = _lib.load_library()
        if not torch.cuda.is_available():
            raise _lib.SvgpError(f"{type(self).__name__} needs a HIP device; there is no CPU execution path")
        self.dev = torch.device(device)
        self.B, self.T, self.px, self.py, self.P, self.H = batch, tmax, px, py, px * py, hidden
        self.clip_grad = bool(clip_grad)
        self.stream = torch.cuda.Stream(device=self.dev)
        f64 = dict(dtype=_F64, device=self.dev)
        self.shapes = dict(mlp_param_shapes(px, py, hidden))
        self.shapes.update(extra_shapes)
        n_tot = sum(int(np.prod(s)) for s in self.shapes.values())
        self.theta, self.grad = torch.zeros(n_tot, **f64), torch.zeros(n_tot, **f64)
        self.adam_m, self.adam_v = torch.zeros(n_tot, **f64), torch.zeros(n_tot, **f64)
        self.params, self.grads, off = {}, {}, 0
        for k, s in self.shapes.items():
            n = int(np.prod(s))
            self.params[k], self.grads[k] = self.theta[off:off + n].view(s), self.grad[off:off + n].view(s)
            off += n
        full = dict(truncated_normal_mlp_params(px, py, hidden, seed))
        full.update(init)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))     # zero-fills ran on torch's stream
        with torch.cuda.stream(self.stream):
            for k, s in self.shapes.items():
                v = full[k]
                self.params[k].copy_(torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v.detach().cpu(),
                                                     dtype=_F64).reshape(s))
        self.state = torch.zeros(STATE_LEN, **f64)
        self.out = torch.zeros(len(OUT_ROWS), batch, **f64)
        self._gemm_scratch = torch.empty(0, **f64)
        self.part = torch.zeros(int(self.lib.svgp_act_bwd_bias_scratch_elems(max(self.P, hidden))), **f64)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        st = torch.zeros(STATE_LEN, dtype=_F64)
        st[STATE["LR"]], st[STATE["BETA"]], st[STATE["LAGRANGE"]] = lr, beta, 1.0
        with torch.cuda.stream(self.stream):
            self.state.copy_(st)
        self.stream.synchronize()
        self.act = {}

    def _gemm(self, ta, tb, M, N, K, A, lda, Bm, ldb, Cm, ldc):
        """Dense-layer GEMM; few output tiles + long contraction -> split-K (svgp_dgemm_splitk)."""
        need = int(self.lib.svgp_dgemm_splitk_scratch_elems(M, N, K))
        if self._gemm_scratch.numel() < need:                           # grows once, at the first step
            self._gemm_scratch = torch.empty(need, dtype=_F64, device=self.dev)
        call("svgp_dgemm_splitk", ta, tb, M, N, K, 1.0, A.data_ptr(), lda, Bm.data_ptr(), ldb, 0.0, Cm.data_ptr(), ldc,
             self._gemm_scratch.data_ptr(), self._gemm_scratch.numel(), self.stream.cuda_stream)

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

    # ---- pieces of a step; all run inside `with torch.cuda.stream(self.stream)`
    def _encode(self, X):
        """(R,P) frames -> h1 (R,H) post-tanh, h2 (R,4) pre-bias head input (VAE_utils.py:26-49)."""
        R, P, H, p, s = X.shape[0], self.P, self.H, self.params, self.stream.cuda_stream
        f64 = dict(dtype=_F64, device=self.dev)
        h1 = torch.empty(R, H, **f64)
        self._gemm(0, 0, R, H, P, X, P, p["encW1"], H, h1, H)
        call("svgp_bias_act_fwd", R, H, 1, p["encB1"].data_ptr(), h1.data_ptr(), s)
        h2 = torch.empty(R, 4, **f64)
        self._gemm(0, 0, R, 4, H, h1, H, p["encW2"], 4, h2, 4)
        return h1, h2

    def _decode_recon(self, z, X, want_grad, row_weights=None):
        """z (R,2) -> logits, pred = sigmoid, per-frame reconstruction term, dlogits (VAE_utils.py:73-94,
        SVGPVAE_model.py:697-700).  row_weights (R): frames whose reconstruction term counts (NP targets)."""
        R, P, H, p, s = z.shape[0], self.P, self.H, self.params, self.stream.cuda_stream
        f64 = dict(dtype=_F64, device=self.dev)
        g1 = torch.empty(R, H, **f64)
        self._gemm(0, 0, R, H, 2, z, 2, p["decW1"], H, g1, H)
        call("svgp_bias_act_fwd", R, H, 1, p["decB1"].data_ptr(), g1.data_ptr(), s)
        logits = torch.empty(R, P, **f64)
        self._gemm(0, 0, R, P, H, g1, H, p["decW2"], P, logits, P)
        call("svgp_bias_act_fwd", R, P, 0, p["decB2"].data_ptr(), logits.data_ptr(), s)
        pred, row_recon = torch.empty(R, P, **f64), torch.empty(R, **f64)
        dlog = torch.empty(R, P, **f64) if want_grad else None
        call("svgp_sigmoid_xent", R, P, 1.0 / self.B, logits.data_ptr(), X.data_ptr(), pred.data_ptr(),
             row_recon.data_ptr(), None if dlog is None else dlog.data_ptr(), s)
        if dlog is not None and row_weights is not None:
            call("svgp_scale_rows", R, P, row_weights.data_ptr(), dlog.data_ptr(), s)
        return g1, pred, row_recon, dlog

    def _decoder_backward(self, z, g1, dlog):
        """-> dz (R,2); writes the decoder gradients."""
        R, P, H, p, g, s = z.shape[0], self.P, self.H, self.params, self.grads, self.stream.cuda_stream
        f64 = dict(dtype=_F64, device=self.dev)
        self._gemm(1, 0, H, P, R, g1, H, dlog, P, g["decW2"], P)
        call("svgp_act_bwd_bias", R, P, 0, None, dlog.data_ptr(), self.part.data_ptr(), g["decB2"].data_ptr(), s)
        dg1 = torch.empty(R, H, **f64)
        self._gemm(0, 1, R, H, P, dlog, P, p["decW2"], P, dg1, H)
        call("svgp_act_bwd_bias", R, H, 1, g1.data_ptr(), dg1.data_ptr(), self.part.data_ptr(), g["decB1"].data_ptr(), s)
        self._gemm(1, 0, 2, H, R, z, 2, dg1, H, g["decW1"], H)
        dz = torch.empty(R, 2, **f64)
        self._gemm(0, 1, R, 2, H, dg1, H, p["decW1"], H, dz, 2)
        return dz

    def _encoder_backward(self, X, h1, dh2):
        R, P, H, p, g, s = X.shape[0], self.P, self.H, self.params, self.grads, self.stream.cuda_stream
        f64 = dict(dtype=_F64, device=self.dev)
        self._gemm(1, 0, H, 4, R, h1, H, dh2, 4, g["encW2"], 4)
        call("svgp_act_bwd_bias", R, 4, 0, None, dh2.data_ptr(), self.part.data_ptr(), g["encB2"].data_ptr(), s)
        dh1 = torch.empty(R, H, **f64)
        self._gemm(0, 1, R, H, 4, dh2, 4, p["encW2"], 4, dh1, H)
        call("svgp_act_bwd_bias", R, H, 1, h1.data_ptr(), dh1.data_ptr(), self.part.data_ptr(), g["encB1"].data_ptr(), s)
        self._gemm(1, 0, P, H, R, X, P, dh1, H, g["encW1"], H)

    def _clip_and_adam(self, adam):
        s = self.stream.cuda_stream
        if self.clip_grad:                                              # BALL_experiment.py:125-127
            call("svgp_clip_by_value", self.grad.numel(), 100000.0, self.grad.data_ptr(), s)
        if adam:
            call("svgp_adam_tf1_step", self.theta.numel(), self.theta.data_ptr(), self.grad.data_ptr(),
                 self.adam_m.data_ptr(), self.adam_v.data_ptr(), self.state.data_ptr(), 0.9, 0.999, 1e-8, s)


class BallStepEngine(_BallMlpEngine):
    """Buffers + kernel schedule of one moving-ball training step (SVGPVAE_Hensman / SVGPVAE_Titsias)."""

    def __init__(self, svgp_x, svgp_y, *, batch=35, tmax=30, px=32, py=32, hidden=500, clip_qs=False, beta=1.0,
                 lr=1e-3, clip_grad=False, device="cuda:0", params=None, seed=0):
        if svgp_x.titsias != svgp_y.titsias or svgp_x.num_inducing_points != svgp_y.num_inducing_points:
            raise ValueError("svgp_x and svgp_y must agree on the ELBO branch and on the number of inducing points")
        if svgp_x.jitter != svgp_y.jitter:
            raise ValueError("svgp_x and svgp_y must use the same jitter")
        self.m, self.titsias = svgp_x.num_inducing_points, svgp_x.titsias
        self.svgp = (svgp_x, svgp_y)
        self.clip_qs = bool(clip_qs)
        # ---- one GP workspace per latent coordinate: rows = frames, channels = videos (validated before any allocation)
        self.cfg = MnistCfg(b=tmax, b_global=tmax, m=self.m, L=batch, M=1, n_obj=0, normalize_obj=0, clip_qs=0, geco=0,
                            train_ip=1, train_gp=1, train_ov=0, b_cap=tmax, clip_pv=2, n_pix=px * py,
                            titsias=int(self.titsias), kl_form=1, single_stat_block=0, N_train=float(tmax),
                            jitter=svgp_x.jitter, kappa_squared=0.0, alpha=0.0, rep_weight=1.0)
        _lib.load_library()
        self.wl = WsLayout()
        call("svgp_mnist_ws_layout_get", C.byref(self.cfg), C.byref(self.wl))
        init = dict(params or {})
        for c, sv in zip("xy", self.svgp):
            init.setdefault(f"ip_{c}", sv.inducing_index_points)
            init.setdefault(f"l_{c}", sv.l_GP)
        self._init_common(dict(ip_x=(self.m,), l_x=(1,), ip_y=(self.m,), l_y=(1,)), init, batch=batch, tmax=tmax, px=px,
                          py=py, hidden=hidden, beta=beta, lr=lr, clip_grad=clip_grad, device=device, seed=seed)
        for c, sv in zip("xy", self.svgp):
            sv.inducing_index_points, sv.l_GP = self.params[f"ip_{c}"], self.params[f"l_{c}"]
        f64 = dict(dtype=_F64, device=self.dev)
        self.ws = [torch.zeros(self.wl.total, **f64) for _ in range(2)]
        self.times = torch.arange(1, tmax + 1, **f64)                   # SVGPVAE_model.py:663
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        self.stream.synchronize()

    def _v(self, c, name, shape):
        off = getattr(self.wl, name)
        return self.ws[c][off:off + int(np.prod(shape))].view(shape)

    # ------------------------------------------------------------------ one step
    def step(self, vid_batch, epsilon=None, adam=True, backward=True):
        """vid_batch (batch,tmax,px,py) float64 CUDA tensor; epsilon (batch,tmax,2) or None (on-device N(0,1)).
        Forward, reverse, optional gradient clip, TF1 Adam when `adam`, per-video ELBO terms and their means."""
        B, T, P, m = self.B, self.T, self.P, self.m
        assert tuple(vid_batch.shape) == (B, T, self.px, self.py)
        p, g, s = self.params, self.grads, self.stream.cuda_stream
        cp, st = C.byref(self.cfg), self.state.data_ptr()
        f64 = dict(dtype=_F64, device=self.dev)
        R = B * T
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.stream):
            X = vid_batch.to(self.dev, _F64).contiguous().view(R, P)
            h1, h2 = self._encode(X)
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
            g1, pred, row_recon, dlog = self._decode_recon(z, X, backward)
            self.act = dict(pred=pred.view(B, T, self.px, self.py), z=z.view(B, T, 2))
            if backward:
                dz = self._decoder_backward(z, g1, dlog)
                call("svgp_ball_unpack_zbar", B, T, dz.data_ptr(), self._v(0, "zbar", (1,)).data_ptr(),
                     self._v(1, "zbar", (1,)).data_ptr(), s)
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
                dh2 = torch.empty(R, 4, **f64)
                yb, sb = fld("ybar"), fld("s2bar")
                call("svgp_ball_head_bwd", B, T, int(self.clip_qs), var_raw[0].data_ptr(), yb[0].data_ptr(),
                     sb[0].data_ptr(), var_raw[1].data_ptr(), yb[1].data_ptr(), sb[1].data_ptr(), dh2.data_ptr(), s)
                self._encoder_backward(X, h1, dh2)
                self._clip_and_adam(adam)
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
            cov.append(Knn - _mm(_mm(Kn, Ki), Kn, tb=True) + _mm(_mm(Kn, Si.mean(0)), Kn, tb=True))
        return (o["elbo"], o["recon"], o["KL_term"], o["inside_elbo"], o["ce_term"], full_p_mu, full_p_var, qnet_mu,
                qnet_var, self.act["pred"], p["l_x"][0].clone(), p["l_y"][0].clone(), o["inside_elbo_recon"],
                o["inside_elbo_kl"], p["ip_x"].clone(), p["ip_y"].clone(), cov[0], cov[1], self)


class PearceStepEngine(_BallMlpEngine):
    """BALL_experiment.py --elbo GPVAE_Pearce | VAE | NP: exact per-video GP regression on the recognition network's
    outputs (build_pearce_elbo_graphs, GPVAE_Pearce_model.py:89-236)."""

    def __init__(self, type_elbo="GPVAE_Pearce", lt=5.0, context_ratio=0.5, GP_joint=False, GP_init=2.0, *, batch=35,
                 tmax=30, px=32, py=32, hidden=500, beta=1.0, lr=1e-3, clip_grad=False, device="cuda:0", params=None,
                 seed=0):
        if type_elbo not in ("GPVAE_Pearce", "VAE", "NP"):
            raise ValueError(f"type_elbo {type_elbo!r}")
        if tmax > 64:
            raise _lib.SvgpError("the exact per-video GP keeps its tmax x tmax matrices in LDS (tmax <= 64)")
        self.type_elbo, self.lt, self.context_ratio, self.GP_joint = type_elbo, float(lt), float(context_ratio), bool(GP_joint)
        l0 = float(GP_init) if GP_joint else float(lt)                  # GPVAE_Pearce_model.py:35-41
        init = dict(params or {})
        init.setdefault("l_x", torch.tensor([l0], dtype=_F64))
        init.setdefault("l_y", torch.tensor([l0], dtype=_F64))
        self._init_common(dict(l_x=(1,), l_y=(1,)), init, batch=batch, tmax=tmax, px=px, py=py, hidden=hidden, beta=beta,
                          lr=lr, clip_grad=clip_grad, device=device, seed=seed)
        f64 = dict(dtype=_F64, device=self.dev)
        B, T = batch, tmax
        self.times = torch.arange(0, tmax, **f64)                       # GPVAE_Pearce_model.py:119-120 (0-based)
        self.buf = {k: torch.zeros(2, T, B, **f64) for k in ("y", "var_raw", "s2", "p_m", "p_v", "eps", "z", "zbar", "ybar",
                                                            "s2bar", "row_ce")}
        self.Ai, self.alpha = torch.zeros(2, B, T, T, **f64), torch.zeros(2, B, T, **f64)
        self.lh, self.ce, self.dl_part = torch.zeros(2, B, **f64), torch.zeros(2, B, **f64), torch.zeros(2, B, **f64)
        # neural-process context sets
        self.c_Ai, self.c_alpha = torch.zeros(2, B, T, T, **f64), torch.zeros(2, B, T, **f64)
        self.c_lh, self.c_dl, self.c_ls = torch.zeros(2, B, **f64), torch.zeros(2, **f64), torch.full((2,), float(lt), **f64)
        self.np_gen = np.random.RandomState(seed + 1)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        self.stream.synchronize()

    def _bufs(self, n=None, idx=None, tmask=None, context=False):
        from ._lib import PearceBufs
        b, p = self.buf, self.params
        q = PearceBufs(B=self.B, T=self.T, n=self.T if n is None else n)
        q.times, q.idx, q.tmask = self.times.data_ptr(), None if idx is None else idx.data_ptr(), \
            None if tmask is None else tmask.data_ptr()
        # the context likelihoods use the constant model length scale (GPVAE_Pearce_model.py:152-153: lt=lt, GP_joint off)
        q.ls_x, q.ls_y = (self.c_ls[0:].data_ptr(), self.c_ls[1:].data_ptr()) if context else \
            (p["l_x"].data_ptr(), p["l_y"].data_ptr())
        for k in ("y", "s2", "p_m", "p_v", "eps", "z", "zbar", "ybar", "s2bar"):
            setattr(q, k + "_x", b[k][0].data_ptr()); setattr(q, k + "_y", b[k][1].data_ptr())
        q.Ai, q.alpha, q.lh = (self.c_Ai if context else self.Ai).data_ptr(), (self.c_alpha if context else self.alpha).data_ptr(), \
            (self.c_lh if context else self.lh).data_ptr()
        q.ce, q.row_ce, q.dl_part = self.ce.data_ptr(), b["row_ce"].data_ptr(), self.dl_part.data_ptr()
        return q

    def draw_context_split(self):
        """GPVAE_Pearce_model.py:121-137: number of context frames ~ round(N(ratio tmax, ratio (1-ratio) tmax)) clipped
        to [2, tmax-2]; an independent random permutation of the frames per video."""
        T, r = self.T, self.context_ratio
        con_tf = int(np.round(min(max(self.np_gen.normal(r * T, math.sqrt(r * (1 - r) * T)), 2), T - 2)))
        ran_ind = np.stack([self.np_gen.permutation(T) for _ in range(self.B)])
        return ran_ind, con_tf

    def step(self, vid_batch, epsilon=None, adam=True, backward=True, ran_ind=None, con_tf=None):
        """One step.  NP: `ran_ind` (batch,tmax) permutations and `con_tf` are drawn when not given."""
        B, T, P = self.B, self.T, self.P
        assert tuple(vid_batch.shape) == (B, T, self.px, self.py)
        p, g, s, b = self.params, self.grads, self.stream.cuda_stream, self.buf
        st = self.state.data_ptr()
        f64 = dict(dtype=_F64, device=self.dev)
        R = B * T
        is_np = self.type_elbo == "NP"
        idx = tmask = None
        if is_np:
            if ran_ind is None:
                ran_ind, con_tf = self.draw_context_split()
            ran_ind = np.asarray(ran_ind)
            idx = torch.as_tensor(ran_ind[:, :con_tf].astype(np.int32)).contiguous()
            tm = np.zeros((B, T))
            np.put_along_axis(tm, ran_ind[:, con_tf:], 1.0, axis=1)
            tmask = torch.as_tensor(tm, dtype=_F64)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.stream):
            if is_np:
                idx, tmask = idx.to(self.dev), tmask.to(self.dev)
            X = vid_batch.to(self.dev, _F64).contiguous().view(R, P)
            h1, h2 = self._encode(X)
            call("svgp_ball_head_fwd", B, T, 0, p["encB2"].data_ptr(), h2.data_ptr(), b["y"][0].data_ptr(),
                 b["var_raw"][0].data_ptr(), b["s2"][0].data_ptr(), b["y"][1].data_ptr(), b["var_raw"][1].data_ptr(),
                 b["s2"][1].data_ptr(), s)
            ex = ey = None
            if epsilon is not None:
                e = epsilon.to(self.dev, _F64)
                ex, ey = e[:, :, 0].t().contiguous(), e[:, :, 1].t().contiguous()
            full = self._bufs(tmask=tmask)
            call("svgp_pearce_gp_fwd", C.byref(full), None if ex is None else ex.data_ptr(),
                 None if ey is None else ey.data_ptr(), st, s)
            ctx = None
            if is_np:
                ctx = self._bufs(n=con_tf, idx=idx, context=True)
                call("svgp_pearce_gp_fwd", C.byref(ctx), None, None, st, s)
            z = torch.empty(R, 2, **f64)
            call("svgp_ball_pack_z", B, T, b["z"][0].data_ptr(), b["z"][1].data_ptr(), z.data_ptr(), s)
            g1, pred, row_recon, dlog = self._decode_recon(z, X, backward, None if tmask is None else tmask.view(-1))
            self.act = dict(pred=pred.view(B, T, self.px, self.py), z=z.view(B, T, 2))
            if backward:
                dz = self._decoder_backward(z, g1, dlog)
                call("svgp_ball_unpack_zbar", B, T, dz.data_ptr(), b["zbar"][0].data_ptr(), b["zbar"][1].data_ptr(), s)
                call("svgp_pearce_gp_bwd", C.byref(full), 1.0, 0, st, g["l_x"].data_ptr(), g["l_y"].data_ptr(), s)
                if is_np:
                    call("svgp_pearce_gp_bwd", C.byref(ctx), -1.0, 1, st, self.c_dl[0:].data_ptr(), self.c_dl[1:].data_ptr(), s)
                if not self.GP_joint:
                    g["l_x"].zero_(); g["l_y"].zero_()
                dh2 = torch.empty(R, 4, **f64)
                call("svgp_ball_head_bwd", B, T, 0, b["var_raw"][0].data_ptr(), b["ybar"][0].data_ptr(),
                     b["s2bar"][0].data_ptr(), b["var_raw"][1].data_ptr(), b["ybar"][1].data_ptr(),
                     b["s2bar"][1].data_ptr(), dh2.data_ptr(), s)
                self._encoder_backward(X, h1, dh2)
                self._clip_and_adam(adam)
            call("svgp_pearce_elbo_assemble", B, T, self.lh.data_ptr(), self.ce.data_ptr(),
                 self.c_lh.data_ptr() if is_np else None, row_recon.data_ptr(), b["row_ce"].data_ptr(),
                 None if tmask is None else tmask.data_ptr(), st, self.out.data_ptr(), s)
            call("svgp_ball_finalize", B, int(bool(adam and backward)), 1, self.out.data_ptr(), st, s)
            self._keep = (idx, tmask, ex, ey)
        return self

    train_step = step

    def outputs(self):
        """build_pearce_elbo_graphs' return tuple (elbo, elbo_recon, elbo_prior_kl, full_p_mu, full_p_var, qnet_mu, qnet_var,
        pred_vid, l_GP_x, l_GP_y, engine)."""
        self.stream.synchronize()
        b = self.buf
        st2 = lambda k: torch.stack([b[k][0].t(), b[k][1].t()], 2).contiguous()
        return (self.out[0].clone(), self.out[1].clone(), self.out[2].clone(), st2("p_m"), st2("p_v"), st2("y"), st2("s2"),
                self.act["pred"], self.params["l_x"][0].clone(), self.params["l_y"][0].clone(), self)


def build_pearce_elbo_graphs(vid_batch, beta, type_elbo="GPVAE_Pearce", lt=5, context_ratio=0.5, GP_joint=False,
                             GP_init=2.0, epsilon=None, params=None, engine=None, ran_ind=None, con_tf=None):
    """GPVAE_Pearce_model.py:89-236: one forward pass on the HIP library; returns the reference's 11-tuple (last slot =
    the engine instead of globals())."""
    B, T, px, py = vid_batch.shape
    eng = engine
    if eng is None:
        hidden = 500 if params is None else int(np.asarray(params["encB1"]).size)
        eng = PearceStepEngine(type_elbo, lt, context_ratio, GP_joint, GP_init, batch=B, tmax=T, px=px, py=py,
                               hidden=hidden, beta=float(beta), params=params)
    eng.set_scalars(beta=float(beta))
    eng.step(vid_batch, epsilon, adam=False, backward=False, ran_ind=ran_ind, con_tf=con_tf)
    return eng.outputs()


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
