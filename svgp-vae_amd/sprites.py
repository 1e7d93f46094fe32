"""SPRITES SVGPVAE_Hensman step on the HIP library (BASELINE config 4 shape).

Reference call surface mirrored here (eager float64 CUDA tensors instead of TF graph tensors):
  spritesVAE(L, im_width=64, im_height=64, n_channels=3)                          VAE_utils.py:275-360
  sprites_representation_network(L, ...)                                           VAE_utils.py:363-391
  spritesSVGP(titsias, fixed_inducing_points, initial_inducing_points, name, jitter, N_train, L_action,
              initial_GPLVM_action, L_character, L, fixed_GP_params, fixed_GPLVM, K_obj_normalize, K_SE)
                                                                                   SVGPVAE_model.py:489-548
  aux_data_sprites_utils(batch_size, N, repeats)                                   SPRITES_utils.py:317-332
  forward_pass_SVGPVAE(..., repr_NN=..., segment_ids=..., repeats=...) -> 16-tuple SVGPVAE_model.py:823-936
and `SpritesStepEngine.train_step` = the reference's `sess.run([optim_step_joint, ...])`
(SPRITES_experiment.py:386-407) incl. optional element-wise gradient clipping (:234-235).

Every per-pixel / per-row / per-matrix operation is a HIP kernel: tap-table MFMA convolutions
(conv_taps.hip), batched MFMA GEMMs for the dense layers (linalg.hip), the product-kernel matrices and
their VJP (gp_sprites.hip), the sparse-GP stages (gp_kernels.hip / gp_large.hip for m > 64), TF1 Adam.
Precision.  The reference runs SPRITES in float32 end to end (VAE_utils.py:277, SVGPVAE_model.py:516).
SpritesStepEngine(net_dtype=torch.float64, gemm_f32=0) computes everything in float64 (a superset).  With
net_dtype=torch.float32 the three networks run in float32 (activations, weights cast from the float64 master copy each
step, gradients cast back; tap-table convolutions and dense layers on v_mfma_f32_16x16x4_f32); gemm_f32=1 additionally
runs every product of the large-m GP block on the float32 MFMA (float64 storage), gemm_f32=2 only the statistics
products.  The m x m factorisations / inverses, the scalar epilogue, Adam and the master parameters stay float64.
"""
import ctypes as C
import os
import math

import numpy as np
import torch

from . import _lib
from ._lib import STATE, STATE_LEN, MnistCfg, SpritesKcfg, WsLayout, call
from .conv import ConvLayer, DeferredSums
from .engine import ExchangeOp, SymBlock, concurrent_streams, dp_pack_enabled

_F64 = torch.float64
ENC_STRIDES = (1, 2, 1, 2, 1, 2)
DEC_UP = (True, False, True, False, True, False, False)


def aux_data_sprites_utils(batch_size, N, repeats):
    """SPRITES_utils.py:317-332."""
    n_char = int(batch_size / N)
    return np.array([[i] * N for i in range(n_char)]).reshape(-1), [repeats for _ in range(n_char)]


def sprites_param_shapes(L, L_character=16):
    """name -> shape in flat-vector order: spritesVAE encoder, decoder (VAE_utils.py:294-338), repr net (:375-391)."""
    shp = []
    for i in range(1, 7):
        shp += [(f"enc_c{i}_w", (3, 3, 3 if i == 1 else 16, 16)), (f"enc_c{i}_b", (16,))]
    shp += [("enc_d_w", (1024, 2 * L)), ("enc_d_b", (2 * L,)), ("dec_d_w", (L, 1024)), ("dec_d_b", (1024,))]
    for i in range(1, 8):
        co = 16 if i < 7 else 3
        shp += [(f"dec_c{i}_w", (3, 3, 16, co)), (f"dec_c{i}_b", (co,))]
    for i, cin in ((1, 3), (2, L_character), (3, L_character)):
        shp += [(f"repr_c{i}_w", (2, 2, cin, L_character)), (f"repr_c{i}_b", (L_character,))]
    return shp


def glorot_uniform_params(L, L_character=16, seed=0):
    rng = np.random.RandomState(seed)
    out = {}
    for name, shp in sprites_param_shapes(L, L_character):
        if name.endswith("_b"):
            out[name] = np.zeros(shp)
            continue
        rf = shp[0] * shp[1] if len(shp) == 4 else 1
        fi, fo = (rf * shp[2], rf * shp[3]) if len(shp) == 4 else shp
        lim = math.sqrt(6.0 / (fi + fo))
        out[name] = rng.uniform(-lim, lim, size=shp)
    return out


class spritesVAE:
    dtype = torch.float64

    def __init__(self, L, im_width=64, im_height=64, n_channels=3, seed=0):
        if (im_width, im_height, n_channels) != (64, 64, 3):
            raise NotImplementedError("spritesVAE is specialised to 64x64x3 frames")
        self.L = L
        self.seed = seed


class sprites_representation_network:
    dtype = torch.float64

    def __init__(self, L, im_width=64, im_height=64, n_channels=3):
        self.L = L            # L_character


class spritesSVGP:
    def __init__(self, titsias, fixed_inducing_points, initial_inducing_points, name, jitter, N_train, L_action,
                 initial_GPLVM_action, L_character, L, fixed_GP_params=False, fixed_GPLVM=False,
                 K_obj_normalize=False, K_SE=False):
        self.titsias, self.jitter, self.N_train, self.L = bool(titsias), float(jitter), float(N_train), L
        self.L_action, self.L_character = L_action, L_character
        self.fixed_inducing_points, self.fixed_GP_params, self.fixed_GPLVM = \
            bool(fixed_inducing_points), bool(fixed_GP_params), bool(fixed_GPLVM)
        self.K_obj_normalize, self.K_SE = bool(K_obj_normalize), bool(K_SE)
        self.inducing_index_points = torch.as_tensor(np.asarray(initial_inducing_points), dtype=_F64).clone()
        self.GPLVM_action = torch.as_tensor(np.asarray(initial_GPLVM_action), dtype=_F64).clone()
        self.nr_inducing = self.inducing_index_points.shape[0]
        # SVGPVAE_model.py:531-540
        self.se = torch.tensor([1.0, 0.1, 1.0, 0.1], dtype=_F64)   # l_action, sigma_action, l_character, sigma_character

    def variable_summary(self):
        return self.GPLVM_action, self.inducing_index_points


class SpritesStepEngine:
    """Buffers + kernel schedule of one rank's SPRITES step."""

    def __init__(self, vae, repr_nn, svgp, *, b_max, seg_len=50, clip_qs=False, geco=False, kappa_squared=0.0075,
                 alpha=0.99, beta=0.001, lr=1e-3, clip_grad=None, device="cuda:0", params=None, rank=0, world_size=1,
                 comm=None, net_dtype=torch.float64, gemm_f32=0, channel_shard=None):
        self.lib = _lib.load_library()
        if not torch.cuda.is_available():
            raise _lib.SvgpError("SpritesStepEngine needs a HIP device; there is no CPU execution path")
        self.dev = torch.device(device)
        self.L, self.La, self.Lc, self.m = vae.L, svgp.L_action, svgp.L_character, svgp.nr_inducing
        self.n_act = svgp.GPLVM_action.shape[0]
        self.seg_len, self.clip_grad, self.geco, self.clip_qs = seg_len, clip_grad, bool(geco), bool(clip_qs)
        self.svgp, self.b_max = svgp, b_max
        # data parallelism over whole character groups: rank r holds b rows of a b_global batch;
        # `comm` (engine.RcclComm) sums the three exchange blocks on the compute stream
        self.rank, self.world_size, self.comm = rank, world_size, comm
        self.stream = torch.cuda.Stream(device=self.dev)
        # side stream of the deferred forward-factor tail (None: everything on the one stream; SVGP_SIDE_STREAMS=0) and a third
        # stream: the first part of the early reverse half runs beside the forward tail (SVGP_SIDE_STREAMS=2: behind it).
        # Measured on one box, m = 800: one stream 31.8 ms, two 30.4, three 30.1 per step (round 2).  The two are picked by the
        # concurrency probe (engine.concurrent_streams): streams that share a hardware queue with self.stream hide nothing.
        mode = os.environ.get("SVGP_SIDE_STREAMS")
        picked = [] if mode == "0" else concurrent_streams(self.stream, 1 if mode == "2" else 2, self.dev)
        self.side = picked[0] if picked else None
        self.side2 = picked[1] if len(picked) > 1 else None
        call("svgp_side_streams_prepare", self.stream.cuda_stream)     # the library's own branches (Cholesky look-ahead) of the three
        for sx in picked:
            call("svgp_side_streams_prepare", sx.cuda_stream)
        f64 = dict(dtype=_F64, device=self.dev)
        assert net_dtype in (torch.float64, torch.float32) and gemm_f32 in (0, 1, 2)
        self.ndt, self.f32 = net_dtype, net_dtype == torch.float32
        self.sfx = "_f32" if self.f32 else ""
        self.dtype_name = ("f32 networks" if self.f32 else "f64 networks") + \
            {0: " + f64 GP block", 1: " + f32-MFMA GP products (f64 factorisations)", 2: " + f32-MFMA GP statistics"}[gemm_f32]
        # ---- flat parameter vector: networks, inducing points, GPLVM table, SE hyper-parameters
        self.shapes = dict(sprites_param_shapes(self.L, self.Lc))
        self.shapes.update(inducing_index_points=(self.m, self.La + self.Lc), GPLVM_action=(self.n_act, self.La), se=(4,))
        n_tot = sum(int(np.prod(s)) for s in self.shapes.values())
        self.theta, self.grad = torch.zeros(n_tot, **f64), torch.zeros(n_tot, **f64)
        self.adam_m, self.adam_v = torch.zeros(n_tot, **f64), torch.zeros(n_tot, **f64)
        self.params, self.grads, off = {}, {}, 0
        self.n_net = sum(int(np.prod(s)) for _, s in sprites_param_shapes(self.L, self.Lc))   # networks = vector prefix
        # networks' view of the parameters / gradients: the float64 vectors themselves, or float32 copies
        self.theta_n = torch.zeros(self.n_net, dtype=self.ndt, device=self.dev) if self.f32 else self.theta
        self.grad_n = torch.zeros(self.n_net, dtype=self.ndt, device=self.dev) if self.f32 else self.grad
        self.np_, self.ng = {}, {}
        for k, s in self.shapes.items():
            n = int(np.prod(s))
            self.params[k] = self.theta[off:off + n].view(s)
            self.grads[k] = self.grad[off:off + n].view(s)
            if off < self.n_net:
                self.np_[k] = self.theta_n[off:off + n].view(s)
                self.ng[k] = self.grad_n[off:off + n].view(s)
            off += n
        init = glorot_uniform_params(self.L, self.Lc, vae.seed) if params is None else params
        init = dict(init)
        # zero-fills run on torch's current stream; the writes below run on self.stream
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        init.setdefault("inducing_index_points", svgp.inducing_index_points)
        init.setdefault("GPLVM_action", svgp.GPLVM_action)
        init.setdefault("se", svgp.se)
        with torch.cuda.stream(self.stream):
            for k, v in init.items():
                self.params[k].copy_(torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v, dtype=_F64)
                                     .reshape(self.shapes[k]))
        svgp.inducing_index_points, svgp.GPLVM_action, svgp.se = \
            self.params["inducing_index_points"], self.params["GPLVM_action"], self.params["se"]
        # ---- GP workspace (shared stage kernels of the MNIST path; model-agnostic fields only)
        # Channel-sharded factor stage: with more than one rank and the large-m path, the (L,m,m) statistics
        # are reduce-SCATTERED over the channels, every rank factors L / G channels and the row stage's inputs are
        # all-gathered -- instead of all-reducing the blocks and factoring all L channels on every rank.
        if channel_shard is None:
            channel_shard = world_size > 1 and self.m > 64 and self.L % world_size == 0 and not svgp.titsias
        if channel_shard and (self.m <= 64 or self.L % world_size or svgp.titsias):
            raise _lib.SvgpError("channel_shard needs m > 64, L divisible by the number of ranks and the Hensman branch")
        self.chan_shard = bool(channel_shard)
        self.base = dict(m=self.m, L=self.L, M=1, n_obj=0, normalize_obj=0, clip_qs=int(clip_qs), geco=int(geco),
                         train_ip=1, train_gp=1, train_ov=0, b_cap=b_max, clip_pv=1, n_pix=64 * 64 * 3,
                         titsias=int(svgp.titsias),
                         N_train=svgp.N_train, jitter=svgp.jitter, kappa_squared=float(kappa_squared),
                         alpha=float(alpha), rep_weight=1.0 if rank == 0 else 0.0,
                         # blocks travel between ranks (also: the 1-rank communicator form of a multi-rank step): one statistics
                         # block per channel, and the workspace carries the wire buffer of the packed exchange
                         single_stat_block=int(self.m > 64 and (world_size > 1 or self.chan_shard)),
                         gemm_f32=int(gemm_f32))
        if self.chan_shard:
            self.base["rep_weight"] = 1.0        # every rank's Kbar holds its channel window's share (sums in the gradient exchange)
        self.cfg = MnistCfg(b=b_max, b_global=b_max, **self.base)
        self.wl = WsLayout()
        call("svgp_mnist_ws_layout_get", C.byref(self.cfg), C.byref(self.wl))
        self.ws = torch.zeros(self.wl.total, **f64)
        self.state = torch.zeros(STATE_LEN, **f64)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        st = torch.zeros(STATE_LEN, dtype=_F64)
        st[STATE["LAGRANGE"]], st[STATE["ALPHA"]] = 1.0, (0.0 if geco else alpha)
        st[STATE["LR"]], st[STATE["BETA"]] = lr, beta
        st[STATE["RNG_CTR"]] = float(rank) * 4294967296.0     # per-rank Philox counter range (see engine.reset_state)
        with torch.cuda.stream(self.stream):
            self.state.copy_(st)
        # ---- layers
        self.enc = [ConvLayer(h, ci, 16, 3, s, "same", dtype=self.ndt) for h, ci, s in
                    zip((64, 64, 32, 32, 16, 16), (3, 16, 16, 16, 16, 16), ENC_STRIDES)]
        dec_h = (8, 16, 16, 32, 32, 64, 64)
        self.dec = [ConvLayer(h, 16, 16 if i < 6 else 3, 3, 1, "same", up=u, dtype=self.ndt)
                    for i, (h, u) in enumerate(zip(dec_h, DEC_UP))]
        self.rep = [ConvLayer(h, ci, self.Lc, 2, 2, "same", dtype=self.ndt) for h, ci in zip((64, 32, 16), (3, self.Lc, self.Lc))]
        self.nwg = 1024       # workgroups of the weight-gradient launches (4 per CU: the fused kernel hides its loads by occupancy)
        self.scratch = torch.zeros(max(l.scratch_elems(self.nwg) for l in self.enc + self.dec + self.rep),
                                   dtype=self.ndt, device=self.dev)
        kc_max = SpritesKcfg(b=b_max, m=self.m, La=self.La, Lc=self.Lc, n_act=self.n_act, normalize=0, k_se=0, rep_weight=1.0)
        self.kscratch = torch.zeros(int(_lib.load_library().svgp_sprites_kernel_bwd_scratch_elems(C.byref(kc_max))), **f64)
        # second weight-gradient scratch: the encoder reverse pass runs on the side stream beside the kernel-matrix reverse pass
        # and the representation network's (phases(), end of the step)
        self.scratch2 = None if self.side is None else torch.zeros(max(l.scratch_elems(self.nwg) for l in self.enc),
                                                                   dtype=self.ndt, device=self.dev)
        # the training step defers the partial sums of every layer's weight / bias gradient to ONE launch at its end
        # (svgp_sum_defer_begin / _flush): every layer then needs its own partial-sum region
        self.lscratch = {(g_, i): torch.zeros(l.scratch_elems(self.nwg), dtype=self.ndt, device=self.dev)
                         for g_, ls in (("enc", self.enc), ("dec", self.dec), ("rep", self.rep)) for i, l in enumerate(ls, 1)}
        self.ones = torch.ones(b_max, 1, dtype=self.ndt, device=self.dev)
        self.stream.synchronize()
        self.act = {}

    # ------------------------------------------------------------------ helpers
    trace = None     # bench.py: a list -> (stage name, event) appended at the stage boundaries of phases()

    def _mark(self, name):
        if self.trace is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(self.stream)
            self.trace.append((name, e))

    def _v(self, name, shape):
        off = getattr(self.wl, name)
        return self.ws[off:off + int(np.prod(shape))].view(shape)

    def _gemm(self, ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, Cm, ldc, stream=None):
        assert A.dtype == B.dtype == Cm.dtype
        call("svgp_sgemm_batched" if A.dtype == torch.float32 else "svgp_dgemm_batched", ta, tb, M, N, K, float(alpha),
             A.data_ptr(), lda, 0, B.data_ptr(), ldb, 0, float(beta), Cm.data_ptr(), ldc, 0, 1,
             self.stream.cuda_stream if stream is None else stream)

    def _sync_net_params(self):
        """float32 networks: refresh the float32 copy of the network weights from the float64 master vector."""
        if self.f32:
            call("svgp_cast_f64_f32", self.n_net, self.theta.data_ptr(), self.theta_n.data_ptr(),
                 torch.cuda.current_stream(self.dev).cuda_stream)

    def _to_net(self, x, stream=None):
        """float64 -> the networks' dtype (a HIP cast kernel on `stream`, default torch's current stream; no-op for float64)."""
        if not self.f32:
            return x
        y = torch.empty(x.shape, dtype=torch.float32, device=self.dev)
        call("svgp_cast_f64_f32", x.numel(), x.contiguous().data_ptr(), y.data_ptr(),
             torch.cuda.current_stream(self.dev).cuda_stream if stream is None else stream)
        return y

    def _to_f64(self, x, stream=None, out=None):
        if not self.f32:
            if out is not None:
                out.copy_(x)
                return out
            return x
        y = torch.empty(x.shape, dtype=_F64, device=self.dev) if out is None else out
        call("svgp_cast_f32_f64", x.numel(), x.contiguous().data_ptr(), y.data_ptr(),
             torch.cuda.current_stream(self.dev).cuda_stream if stream is None else stream)
        return y

    def _colsum(self, X, rows, cols, out, stream=None):
        """out (cols) = column sums of X (rows, cols): 1^T X on the library GEMM (bias gradients of the dense layers)."""
        self._gemm(1, 0, 1, cols, rows, 1.0, self.ones, 1, X, cols, 0.0, out, cols, stream=stream)

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

    # ------------------------------------------------------------------ forward pieces (run inside self.stream)
    def _encoder_forward(self, images, b):
        """spritesVAE.encode (VAE_utils.py:294-315,343-349): 6 convs, Dense(2L), exp / clip head."""
        p, s, L = self.np_, self.stream.cuda_stream, self.L
        nd = dict(dtype=self.ndt, device=self.dev)
        a, x = [], images
        for i, lay in enumerate(self.enc, 1):
            out = torch.empty(b, lay.Ho, lay.Ho, lay.Co, **nd)
            lay.forward(x, p[f"enc_c{i}_w"], p[f"enc_c{i}_b"], out, s)
            a.append(out); x = out
        enc = torch.empty(b, 2 * L, **nd)
        self._gemm(0, 0, b, 2 * L, 1024, 1.0, x, 1024, p["enc_d_w"], 2 * L, 0.0, enc, 2 * L)
        enc = self._to_f64(enc)                               # the head (bias, exp, clip) and the GP block are float64
        mu, var_raw, var = self._v("qnet_mu", (b, L)), self._v("qnet_var_raw", (b, L)), self._v("qnet_var", (b, L))
        call("svgp_enc_head_fwd", b, L, int(self.clip_qs), self.params["enc_d_b"].data_ptr(), enc.data_ptr(), mu.data_ptr(),
             var_raw.data_ptr(), var.data_ptr(), s)
        return a, enc, mu, var_raw, var

    def _repr_forward(self, images, b, stream=None):
        """sprites_representation_network (VAE_utils.py:375-391): per-frame character vectors (b, L_character).
        stream: torch stream the launches go to (the caller has made it current); default the engine's."""
        p, s = self.np_, (self.stream if stream is None else stream).cuda_stream
        nd = dict(dtype=self.ndt, device=self.dev)
        r, x = [], images
        for i, lay in enumerate(self.rep, 1):
            out = torch.empty(b, lay.Ho, lay.Ho, lay.Co, **nd)
            lay.forward(x, p[f"repr_c{i}_w"], p[f"repr_c{i}_b"], out, s)
            r.append(out); x = out
        rvec = torch.empty(b, self.Lc, **nd)
        call("svgp_avgpool_fwd" + self.sfx, b, 64, self.Lc, x.data_ptr(), rvec.data_ptr(), s)
        return r, self._to_f64(rvec, stream=s)

    def _decoder_forward(self, z, b):
        """spritesVAE.decode (VAE_utils.py:317-338,352-360): Dense(1024) -> (8,8,16) -> 7 (up)convs."""
        p, s, L = self.np_, self.stream.cuda_stream, self.L
        nd = dict(dtype=self.ndt, device=self.dev)
        z = self._to_net(z)
        h0 = torch.empty(b, 1024, **nd)
        self._gemm(0, 0, b, 1024, L, 1.0, z, L, p["dec_d_w"], 1024, 0.0, h0, 1024)
        call("svgp_bias_add" + self.sfx, b, 1024, p["dec_d_b"].data_ptr(), h0.data_ptr(), s)
        d, x = [], h0.view(b, 8, 8, 16)
        for i, lay in enumerate(self.dec, 1):
            out = torch.empty(b, lay.Ho, lay.Ho, lay.Co, **nd)
            lay.forward(x, p[f"dec_c{i}_w"], p[f"dec_c{i}_b"], out, s)
            d.append(out); x = out
        return h0, d

    # ------------------------------------------------------------------ forward-only entry points (test pipeline)
    def _chunks(self, n):
        return [(lo, min(lo + self.b_max, n)) for lo in range(0, n, self.b_max)]

    def encode(self, images):
        """(qnet_mu, qnet_var) of spritesVAE.encode with the clip of SVGPVAE_model.py:961-962, any number of frames."""
        n = images.shape[0]
        mu_o, var_o = torch.empty(n, self.L, dtype=_F64, device=self.dev), torch.empty(n, self.L, dtype=_F64, device=self.dev)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.stream):
            self._sync_net_params()
            for lo, hi in self._chunks(n):
                _, _, mu, _, var = self._encoder_forward(images[lo:hi].to(self.ndt).contiguous(), hi - lo)
                mu_o[lo:hi].copy_(mu); var_o[lo:hi].copy_(var)
        self.stream.synchronize()
        return mu_o, var_o

    def character_vectors(self, images):
        """repr_nn.repr_nn(frames): (n, L_character)."""
        n = images.shape[0]
        out = torch.empty(n, self.Lc, dtype=_F64, device=self.dev)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.stream):
            self._sync_net_params()
            for lo, hi in self._chunks(n):
                out[lo:hi].copy_(self._repr_forward(images[lo:hi].to(self.ndt).contiguous(), hi - lo)[1])
        self.stream.synchronize()
        return out

    def decode(self, z):
        """spritesVAE.decode for any number of latent rows: (n, 64, 64, 3)."""
        n = z.shape[0]
        out = torch.empty(n, 64, 64, 3, dtype=_F64, device=self.dev)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.stream):
            self._sync_net_params()
            for lo, hi in self._chunks(n):
                out[lo:hi].copy_(self._decoder_forward(z[lo:hi].contiguous(), hi - lo)[1][-1])
        self.stream.synchronize()
        return out

    def kernel_matrices(self, aux):
        """spritesSVGP.kernel_matrix for batch rows `aux` (n, 1 + L_character) against the inducing points:
        (K_mm (m,m), K_nm (n,m), diag K_nn (n)), float64 (svgp_sprites_kernel_matrix_fwd)."""
        n = aux.shape[0]
        p, s = self.params, self.stream.cuda_stream
        K = torch.empty(self.m, self.m, dtype=_F64, device=self.dev)
        Kn = torch.empty(n, self.m, dtype=_F64, device=self.dev)
        knn = torch.empty(n, dtype=_F64, device=self.dev)
        kc = SpritesKcfg(b=n, m=self.m, La=self.La, Lc=self.Lc, n_act=self.n_act,
                         normalize=int(self.svgp.K_obj_normalize), k_se=int(self.svgp.K_SE), rep_weight=1.0)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.stream):
            aux = aux.to(self.dev, _F64).contiguous()
            call("svgp_sprites_kernel_matrix_fwd", C.byref(kc), aux.data_ptr(), p["inducing_index_points"].data_ptr(),
                 p["GPLVM_action"].data_ptr(), p["se"].data_ptr(), K.data_ptr(), Kn.data_ptr(), knn.data_ptr(), s)
        self.stream.synchronize()
        return K, Kn, knn

    # ------------------------------------------------------------------ one step
    def step(self, images, action_ids, eps=None, adam=True, b_global=None):
        """images (b,64,64,3), action_ids (b) float64 CUDA tensors; eps (b,L) or None (on-device N(0,1)).
        Runs forward, reverse, (clip), TF1 Adam when `adam`, and the scalar epilogue.  With world_size > 1
        the three exchange blocks are summed over ranks by `self.comm` between the phases."""
        if self.world_size > 1 and self.comm is None:
            raise _lib.SvgpError("world_size > 1 needs a communicator (engine.RcclComm)")
        if self.world_size > 1 and not getattr(self, "_blocks_checked", False):
            # the exchange blocks' lengths are functions of the row capacity: every rank must have been built with the same b_max
            from .engine import agree_on_lengths
            self._blocks_checked = agree_on_lengths([self.wl.statA_len, self.wl.statB_len, self.wl.gradC_len], self.comm,
                                                    self.dev, self.stream)
        tr = getattr(self, "exchange_trace", None)
        for ops in self.phases(images, action_ids, eps, adam, b_global):
            if self.comm is not None:
                if tr is not None:           # bench.py: events around every exchange point (pack + grouped launch + unpack)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                self.comm.run(ops, self.stream.cuda_stream)
                if tr is not None:
                    e1.record(self.stream)
                    tr.append((e0, e1))
        return self

    def phases(self, images, action_ids, eps=None, adam=True, b_global=None):
        """The step as a generator (see _phases_body).  Whatever ends it -- exhaustion, an exception in a stage, the caller
        abandoning it at an exchange point -- the per-step weight layouts of the layers are dropped, so that a later forward()
        / backward() can never pick up layouts of parameters that have changed in place since."""
        try:
            return (yield from self._phases_body(images, action_ids, eps, adam, b_global))
        finally:
            for lay in self.enc + self.dec + self.rep:
                lay._prep = None

    def _phases_body(self, images, action_ids, eps=None, adam=True, b_global=None):
        """Generator over the step: yields, at every exchange point, the list of engine.ExchangeOp to run across the
        ranks.  Three points (all-reduce of the forward statistics, the backward statistics, gradients + scalar sums) in
        the plain form; five in the channel-sharded form (reduce-scatter S, v | all-gather Sigma^-1, t, u |
        reduce-scatter A2, ud, td | all-gather Ssym, vbar | all-reduce gradients + sums)."""
        b = images.shape[0]
        assert b <= self.b_max and b % self.seg_len == 0
        b_global = b * self.world_size if b_global is None else b_global
        p, g, s, L = self.params, self.grads, self.stream.cuda_stream, self.L
        pn, gn, sfx = self.np_, self.ng, self.sfx            # the networks' (possibly float32) parameters / gradients
        nd = dict(dtype=self.ndt, device=self.dev)
        cfg = MnistCfg(b=b, b_global=b_global, **self.base)
        self.cfg = cfg
        cp = C.byref(cfg)
        ws, st = self.ws.data_ptr(), self.state.data_ptr()
        f64 = dict(dtype=_F64, device=self.dev)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.stream):
            images = images.to(self.ndt).contiguous()
            self._sync_net_params()
            # forward-layout / transposed tap weights of every layer, once per step and OFF the launch chains of the forward and
            # reverse passes: on the third stream (idle until the reverse factor stage) when there is one.  The encoder's and the
            # representation network's forward weights are the parameters themselves, so nothing at the head of the step waits.
            prep_stream = self.side2 if self.side2 is not None else self.stream
            if prep_stream is not self.stream:
                prep_stream.wait_stream(self.stream)
            for grp, layers in (("dec", self.dec), ("enc", self.enc), ("repr", self.rep)):
                for i, lay in enumerate(layers, 1):
                    lay.prepare(pn[f"{grp}_c{i}_w"], prep_stream.cuda_stream, need_bwd=not (grp != "dec" and i == 1))
            prep_done = torch.cuda.Event()
            prep_done.record(prep_stream)      # (an event, not the stream: more work goes to that stream later in the step)
            deferred = DeferredSums()          # every layer's closing partial sums: ONE launch at the end of the step
            self._mark("nets_fwd_enc")
            kc = SpritesKcfg(b=b, m=self.m, La=self.La, Lc=self.Lc, n_act=self.n_act,
                             normalize=int(self.svgp.K_obj_normalize), k_se=int(self.svgp.K_SE),
                             # Kbar-derived terms (m <= 64): replicated on every rank, counted on rank 0.  m > 64: the reverse
                             # factor stage has applied cfg.rep_weight to the replicated part of Kbar and added this rank's
                             # row-local share unweighted (gp_large.hip), so Kbar counts as it is on every rank
                             rep_weight=1.0 if (self.rank == 0 or self.m > 64) else 0.0)
            K, Kn, knn = self._v("K", (self.m, self.m)), self._v("Kn", (b, self.m)), self._v("knn", (b,))
            aid = action_ids.to(_F64).contiguous()

            def repr_and_kernel_matrices(strm):
                r_, rvec = self._repr_forward(images, b, stream=strm)
                aux_ = torch.empty(b, 1 + self.Lc, **f64)
                call("svgp_sprites_aux_fwd", b, self.seg_len, self.Lc, rvec.data_ptr(), aid.data_ptr(), aux_.data_ptr(),
                     strm.cuda_stream)
                call("svgp_sprites_kernel_matrix_fwd", C.byref(kc), aux_.data_ptr(), p["inducing_index_points"].data_ptr(),
                     p["GPLVM_action"].data_ptr(), p["se"].data_ptr(), K.data_ptr(), Kn.data_ptr(), knn.data_ptr(),
                     strm.cuda_stream)
                return r_, aux_

            # the step opens with two independent chains of small launches: representation network -> auxiliary data -> kernel
            # matrices, and the encoder; the first goes to the side stream (SVGP_SPRITES_ENC_SIDE=0 / one stream: in line)
            if self.side is not None and os.environ.get("SVGP_SPRITES_ENC_SIDE") != "0":
                self.side.wait_stream(self.stream)
                with torch.cuda.stream(self.side):
                    r, aux = repr_and_kernel_matrices(self.side)
                a, enc, mu, var_raw, var = self._encoder_forward(images, b)
                self.stream.wait_stream(self.side)
            else:
                a, enc, mu, var_raw, var = self._encoder_forward(images, b)
                r, aux = repr_and_kernel_matrices(self.stream)
            # ---------------- sparse-GP block
            self._mark("gp_fwd_stats")
            call("svgp_gp_stats_fwd", cp, ws, s)
            if self.svgp.titsias:
                call("svgp_gp_titsias_stats", cp, ws, s)
        mm_, G_, r_ = self.m * self.m, self.world_size, self.rank
        nl = L // G_ if self.chan_shard else L
        fld = lambda name, per: self.ws[getattr(self.wl, name):getattr(self.wl, name) + L * per]       # an (L, per) field
        plain = lambda kind, *fields: [ExchangeOp(kind, fld(n_, per)) for n_, per in fields]
        # channel-sharded exchange (the sequence of svgp_mnist_train_step_dp): five points, each ONE grouped RCCL launch; the
        # symmetric (L,m,m) members travel tile-packed (engine.SymBlock) from m >= 512
        pack = self.chan_shard and dp_pack_enabled(self.m)
        pe = int(_lib.load_library().svgp_sym_packed_elems(self.m))
        if pack and self.wl.xpack_len < L * pe:
            raise _lib.SvgpError("the packed exchange needs the workspace's wire buffer (world_size > 1) or SVGP_DP_PACK=0")
        xp = [self.ws[self.wl.xpack:self.wl.xpack + L * pe]] if pack else [None]     # ONE wire buffer per exchange point
        sym = lambda k, avg=False, pre=False: SymBlock(self.m, L, avg, xp[k], pre) if pack else None
        l0 = r_ * nl
        wptr = lambda name, k=None: (xp[k].data_ptr() + 8 * l0 * pe) if k is not None else \
            (self.ws.data_ptr() + 8 * (getattr(self.wl, name) + l0 * mm_))
        if self.chan_shard:
            yield [ExchangeOp("reduce_scatter", fld("S", mm_), sym(0))] + plain("reduce_scatter", ("v", self.m))
        else:
            yield [ExchangeOp("allreduce", self.ws[self.wl.statA:self.wl.statA + self.wl.statA_len])]
        eps_ptr = None if eps is None else eps.contiguous().data_ptr()
        # The side work of the forward factor stage -- its tail ((A_hat + jI)^-1, log det, KL_l: a whole batched inverse that only
        # the reverse factor stage and the final scalars need; include/svgpvae_hip.h: svgp_gp_factor_fwd_aji_tail) and the early
        # half of the reverse factor stage (no reverse statistic needed) -- runs beside the row stage, the decoder and the reverse
        # statistics.  It is FORKED right behind the stage but ISSUED behind the row stage: its launches (two blocked factorisations)
        # take the host ~0.5 ms to enqueue, and the caller's stream must not run dry meanwhile; the branch has the slack.
        side_work = None
        with torch.cuda.stream(self.stream):
            self._mark("gp_fwd_factor")
            if self.chan_shard:
                call("svgp_gp_factor_fwd_channels_part", cp, l0, nl, 1, ws, s)         # without the (A_hat + jI)^-1 tail
                if pack:   # the window in wire format BEFORE the side branch starts reading it
                    call("svgp_sym_pack", self.m, nl, 0, wptr("Si"), wptr("Si", 0), s)
                sd = self.side if self.side is not None else self.stream
                sd.wait_stream(self.stream)

                def side_work():
                    call("svgp_gp_factor_fwd_channels_part", cp, l0, nl, 2, ws, sd.cuda_stream)
                    if self.side is not None:
                        call("svgp_gp_factor_bwd_channels_part", cp, l0, nl, 1, ws, st, sd.cuda_stream)
            elif self.m > 64 and self.side is not None and not self.svgp.titsias:
                # (not with titsias: svgp_gp_titsias_fwd inverts through the same scratch, ws.scr_inv, on the main stream)
                call("svgp_gp_factor_fwd_defer_aji", cp, ws, s)
                self.side.wait_stream(self.stream)
                if self.side2 is not None:
                    self.side2.wait_stream(self.stream)

                def side_work():
                    call("svgp_gp_factor_fwd_aji_tail", cp, ws, self.side.cuda_stream)
                    # the first part of the early reverse half beside the tail on a third stream (it does not need the tail's
                    # inverse), its second part behind both
                    if self.side2 is not None:
                        call("svgp_gp_factor_bwd_early_a", cp, ws, st, self.side2.cuda_stream)
                        self.side.wait_stream(self.side2)
                        call("svgp_gp_factor_bwd_early_b", cp, ws, st, self.side.cuda_stream)
                    else:
                        call("svgp_gp_factor_bwd_early", cp, ws, st, self.side.cuda_stream)
            else:
                call("svgp_gp_factor_fwd", cp, ws, s)
        if self.chan_shard:
            # (round 4: M2 = Ki A Ki is neither formed nor exchanged -- the row stage evaluates k^T M2 k as w^T Si w)
            yield [ExchangeOp("allgather", fld("Si", mm_), sym(0, pre=True))] + plain("allgather", ("t", self.m), ("u", self.m))
        with torch.cuda.stream(self.stream):
            call("svgp_gp_posterior_fwd", cp, eps_ptr, ws, st, s)
            if side_work is not None:
                side_work()
            if self.svgp.titsias:
                call("svgp_gp_titsias_fwd", cp, ws, st, s)
            # ---------------- decoder
            z = self._v("z", (b, L))
            self._mark("nets_fwd_dec")
            if prep_stream is not self.stream:
                self.stream.wait_event(prep_done)
            h0, d = self._decoder_forward(z, b)
            x = d[-1]
            recon = x
            tot = b * 64 * 64 * 3
            call("svgp_sqerr_fwd" + sfx, tot, min(b, 256), images.data_ptr(), recon.data_ptr(),
                 self._v("part_sums", (1,)).data_ptr(), s)
            # ================ reverse
            self._mark("nets_bwd_dec")
            dx = torch.empty_like(recon)
            call("svgp_sqerr_bwd" + sfx, tot, int(self.geco), b_global, 64 * 64 * 3, st, images.data_ptr(), recon.data_ptr(),
                 dx.data_ptr(), s)
            for i in range(7, 0, -1):
                lay = self.dec[i - 1]
                xin = d[i - 2] if i > 1 else h0.view(b, 8, 8, 16)
                dx = lay.backward(xin, pn[f"dec_c{i}_w"], d[i - 1], dx, gn[f"dec_c{i}_w"], gn[f"dec_c{i}_b"],
                                  self.lscratch[("dec", i)], s, nwg=self.nwg, deferred=deferred)
            dh0 = dx.view(b, 1024)
            zn = self._to_net(z)
            self._gemm(1, 0, L, 1024, b, 1.0, zn, L, dh0, 1024, 0.0, gn["dec_d_w"], 1024)
            self._colsum(dh0, b, 1024, gn["dec_d_b"])
            zbar = self._v("zbar", (b, L))
            if self.f32:
                zb = torch.empty(b, L, **nd)
                self._gemm(0, 1, b, L, 1024, 1.0, dh0, 1024, pn["dec_d_w"], 1024, 0.0, zb, L)
                self._to_f64(zb, out=zbar)
            else:
                self._gemm(0, 1, b, L, 1024, 1.0, dh0, 1024, pn["dec_d_w"], 1024, 0.0, zbar, L)
            self._mark("gp_bwd_stats")
            call("svgp_gp_stats_bwd", cp, ws, st, s)
        if self.chan_shard:
            yield [ExchangeOp("reduce_scatter", fld("A2", mm_), sym(0))] + plain("reduce_scatter", ("ud", self.m), ("td", self.m))
        else:
            yield [ExchangeOp("allreduce", self.ws[self.wl.statB:self.wl.statB + self.wl.statB_len])]
        with torch.cuda.stream(self.stream):
            self._mark("gp_bwd_factor")
            if self.chan_shard:
                if self.side is not None:
                    self.stream.wait_stream(self.side)
                call("svgp_gp_factor_bwd_channels_part", cp, l0, nl, 2 if self.side is not None else 0, ws, st, s)
            elif self.m > 64 and self.side is not None and not self.svgp.titsias:
                # the part of the late half that reads nothing the side branch writes -- the vector chain and, with all rows local, X and
                # the two full products Si X, (Si X) Si: 2.3 ms at m = 800 -- BEFORE the join: the caller's stream used to wait 0.8 ms
                # there for the second inverse + early half with that work ready (kernel trace, round 5)
                call("svgp_gp_factor_bwd_late_a", cp, ws, st, s)
                self.stream.wait_stream(self.side)
                if os.environ.get("SVGP_KBAR_BRANCH") != "0":
                    # round 6 (as csrc/api.hip does for the MNIST step): the single-matrix chain of the gradient of Ki on the branch that
                    # has just been joined, beside the channel block
                    self.side.wait_stream(self.stream)
                    call("svgp_gp_factor_bwd_late_b_kbar", cp, ws, st, self.side.cuda_stream)
                    call("svgp_gp_factor_bwd_late_b_channels", cp, ws, st, s)
                    self.stream.wait_stream(self.side)
                    call("svgp_gp_factor_bwd_late_b_final", cp, ws, st, s)
                else:
                    call("svgp_gp_factor_bwd_late_b", cp, ws, st, s)
            else:
                call("svgp_gp_factor_bwd", cp, ws, st, s)
        if self.chan_shard:      # (KL_l comes out of the tail, joined above)
            yield [ExchangeOp("allgather", fld("Ssym", mm_), sym(0))] + plain("allgather", ("vbar", self.m), ("KL", 1))
        with torch.cuda.stream(self.stream):
            call("svgp_gp_posterior_bwd", cp, ws, st, s)
            if self.svgp.titsias:
                call("svgp_gp_titsias_bwd", cp, ws, st, s)
            # The end of the step is two independent chains of small launches (5 - 7 us each, the GPU is waiting for the next one
            # most of the time): kernel-matrix reverse pass -> representation network, and encoder head -> encoder.  Both need
            # only the reverse row stage above, so the encoder chain goes to the side stream (idle since the reverse factor stage)
            # with its own weight-gradient scratch: 1.12 -> 0.65 ms for this tail at m = 800.  SVGP_SIDE_STREAMS=0: in line.
            enc_side = self.side is not None and os.environ.get("SVGP_SPRITES_ENC_SIDE") != "0"

            def encoder_bwd(sx, scratch):
                d_enc = torch.empty(b, 2 * L, **f64)
                call("svgp_enc_head_bwd", b, L, int(self.clip_qs), var_raw.data_ptr(), self._v("ybar", (1,)).data_ptr(),
                     self._v("s2bar", (1,)).data_ptr(), d_enc.data_ptr(), sx)
                d_enc = self._to_net(d_enc, stream=sx)
                a6 = a[5].view(b, 1024)
                self._gemm(1, 0, 1024, 2 * L, b, 1.0, a6, 1024, d_enc, 2 * L, 0.0, gn["enc_d_w"], 2 * L, stream=sx)
                self._colsum(d_enc, b, 2 * L, gn["enc_d_b"], stream=sx)
                dxe = torch.empty(b, 8, 8, 16, **nd)
                self._gemm(0, 1, b, 1024, 2 * L, 1.0, d_enc, 2 * L, pn["enc_d_w"], 2 * L, 0.0, dxe, 1024, stream=sx)
                for i in range(6, 0, -1):
                    xin = a[i - 2] if i > 1 else images
                    dxe = self.enc[i - 1].backward(xin, pn[f"enc_c{i}_w"], a[i - 1], dxe, gn[f"enc_c{i}_w"], gn[f"enc_c{i}_b"],
                                                   self.lscratch[("enc", i)], sx, need_dx=i > 1, nwg=self.nwg, deferred=deferred)

            if enc_side:
                self.side.wait_stream(self.stream)
                with torch.cuda.stream(self.side):
                    encoder_bwd(self.side.cuda_stream, self.scratch2)
            d_char = torch.empty(b, self.Lc, **f64)
            call("svgp_sprites_kernel_matrix_bwd", C.byref(kc), aux.data_ptr(), p["inducing_index_points"].data_ptr(),
                 p["GPLVM_action"].data_ptr(), p["se"].data_ptr(), self._v("Kbar", (1,)).data_ptr(),
                 self._v("Knbar", (1,)).data_ptr(), self._v("knnbar", (1,)).data_ptr(),
                 g["inducing_index_points"].data_ptr(), g["GPLVM_action"].data_ptr(), d_char.data_ptr(),
                 g["se"].data_ptr(), self.kscratch.data_ptr(), self.kscratch.numel(), s)
            self._mark("nets_bwd_enc")
            d_rvec = torch.empty(b, self.Lc, **f64)
            call("svgp_sprites_aux_bwd", b, self.seg_len, self.Lc, d_char.data_ptr(), d_rvec.data_ptr(), s)
            d_rvec = self._to_net(d_rvec)
            dx = torch.empty(b, 8, 8, self.Lc, **nd)
            call("svgp_avgpool_bwd" + sfx, b, 64, self.Lc, d_rvec.data_ptr(), dx.data_ptr(), s)
            for i in range(3, 0, -1):
                xin = r[i - 2] if i > 1 else images
                dx = self.rep[i - 1].backward(xin, pn[f"repr_c{i}_w"], r[i - 1], dx, gn[f"repr_c{i}_w"], gn[f"repr_c{i}_b"],
                                              self.lscratch[("rep", i)], s, need_dx=i > 1, nwg=self.nwg, deferred=deferred)
            if enc_side:
                self.stream.wait_stream(self.side)
            else:
                encoder_bwd(s, self.scratch)
            # every layer's weight / bias partial sums in one launch, then the folds of the up layers' effective-weight gradients;
            # the per-step weight layouts are dropped (the parameters change in the optimiser step below)
            deferred.flush(s)
            for lay in self.enc + self.dec + self.rep:
                lay._prep = None
            if self.f32:                                       # float32 network gradients -> the float64 gradient vector
                call("svgp_cast_f32_f64", self.n_net, self.grad_n.data_ptr(), self.grad.data_ptr(), s)
            # frozen parameter groups (inverted flags of SPRITES_experiment.py:109-111)
            if getattr(self, "freeze_repr", False):           # --repr_nn_pretrain yes_fixed (SPRITES_experiment.py:214-216)
                for k in g:
                    if k.startswith("repr_"):
                        g[k].zero_()
            if self.svgp.fixed_inducing_points:
                g["inducing_index_points"].zero_()
            if self.svgp.fixed_GPLVM:
                g["GPLVM_action"].zero_()
            if self.svgp.fixed_GP_params or not self.svgp.K_SE:
                g["se"].zero_()
            call("svgp_mnist_grad_reduce", cp, ws, s)           # scalar partial sums -> ws.sums
        yield [ExchangeOp("allreduce", self.grad), ExchangeOp("allreduce", self.ws[self.wl.sums:self.wl.sums + 8])]
        with torch.cuda.stream(self.stream):
            self._mark("optim")
            if self.clip_grad is not None:
                call("svgp_clip_by_value", self.grad.numel(), float(self.clip_grad), self.grad.data_ptr(), s)
            if adam:
                call("svgp_adam_tf1_step", self.theta.numel(), self.theta.data_ptr(), self.grad.data_ptr(),
                     self.adam_m.data_ptr(), self.adam_v.data_ptr(), st, 0.9, 0.999, 1e-8, s)
                call("svgp_elbo_finalize", cp, ws, st, s)
            else:
                call("svgp_elbo_finalize_noadam", cp, ws, st, s)
            self._mark("end")
            self.act = dict(recon=recon.double() if self.f32 else recon, aux=aux, enc=enc)     # (outputs(): not on the step's path)

    def outputs(self):
        """The 16-tuple of forward_pass_SVGPVAE for the last step (mean_vectors slot = aux data)."""
        self.stream.synchronize()
        b, L = self.cfg.b, self.L
        sc = self.scalars()
        t = lambda v: torch.tensor(v, dtype=_F64, device=self.dev)
        w = lambda n: self._v(n, (b, L)).clone()
        return (t(sc["elbo"]), t(sc["recon_loss"]), t(sc["kl_term"]), t(sc["inside_elbo"]), t(sc["ce_term"]), w("p_m"),
                w("p_v"), w("qnet_mu"), w("qnet_var"), self.act["recon"], t(sc["inside_recon"]), t(sc["inside_kl"]),
                w("z"), t(sc["c_ma"]), t(sc["lagrange"]), self.act["aux"])


def forward_pass_SVGPVAE(data_batch, beta, vae, svgp, C_ma, lagrange_mult, alpha, kappa, clipping_qs=False, GECO=False,
                         repr_NN=None, segment_ids=None, repeats=None, bias_analysis=False, epsilon=None, params=None,
                         engine=None):
    """SVGPVAE_model.py:823-936 with repr_NN set (SPRITES).  data_batch = (frames (b,64,64,3), action_IDs (b)).
    segment_ids / repeats as produced by aux_data_sprites_utils: equal-length contiguous segments."""
    if repr_NN is None:
        raise ValueError("SPRITES path needs the representation network (use SVGPVAE_model.forward_pass_SVGPVAE for MNIST)")
    frames, action_ids = data_batch
    b = frames.shape[0]
    seg_len = int(repeats[0]) if repeats is not None else b
    eng = engine or getattr(svgp, "_engine", None)
    if eng is None or eng.b_max < b or eng.seg_len != seg_len or eng.geco != bool(GECO) or eng.clip_qs != bool(clipping_qs):
        eng = SpritesStepEngine(vae, repr_NN, svgp, b_max=b, seg_len=seg_len, clip_qs=clipping_qs, geco=GECO,
                                kappa_squared=float(kappa) ** 2, beta=float(beta), params=params)
    _attach(eng, svgp, vae, repr_NN)
    eng.set_scalars(beta=float(beta), c_ma=float(C_ma), lagrange=float(lagrange_mult), alpha=float(alpha))
    dev = eng.dev
    eng.step(frames.to(dev, _F64), action_ids.to(dev, _F64), None if epsilon is None else epsilon.to(dev, _F64),
             adam=False)
    return eng.outputs()


# ---------------------------------------------------------------------------------------------
# conditional generation for a test character (SPRITES_experiment.py:160-205, 364-372, 500-560)
# ---------------------------------------------------------------------------------------------
def general_inverse(A, stream=None):
    """`tf.linalg.inv(A)` of ONE general square float64 matrix on the device: LU with partial pivoting in the library
    (svgp_lu_inverse; SPRITES_experiment.py:178 inverts K_mm WITHOUT jitter, rank-deficient with the linear kernels)."""
    A = A.to(_F64).contiguous()
    m = A.shape[0]
    assert A.ndim == 2 and A.shape[1] == m and A.is_cuda
    lib = _lib.load_library()
    out = torch.empty_like(A)
    work = torch.empty(int(lib.svgp_lu_inverse_workspace_elems(m)), dtype=_F64, device=A.device)
    s = torch.cuda.current_stream(A.device) if stream is None else stream
    call("svgp_lu_inverse", m, A.data_ptr(), out.data_ptr(), work.data_ptr(), s.cuda_stream)
    s.synchronize()                   # (work is freed on return)
    return out


def _attach(eng, *objs):
    """The reference's functions take the network / GP objects only (e.g. aux_data_SVGPVAE_sprites(data_batch, repr_nn,
    segment_ids, repeats), SVGPVAE_model.py:1086); the engine that owns their parameters hangs on each of them."""
    for o in objs:
        if o is not None:
            o._engine = eng
    return eng


def _attached(*objs):
    for o in objs:
        eng = getattr(o, "_engine", None)
        if eng is not None:
            return eng
    return None


def _engine_of(svgp, vae, repr_nn, engine, b_hint, clipping_qs):
    if engine is not None:          # an explicit engine (e.g. the driver's evaluation engine) is used, never attached
        return engine
    eng = _attached(svgp, repr_nn, vae)
    if eng is None:
        if svgp is None:
            raise _lib.SvgpError("no step engine is attached to these network objects yet: run forward_pass_SVGPVAE once, "
                                 "build a SpritesStepEngine(vae, repr_nn, svgp, ...), or pass svgp= / engine=")
        eng = SpritesStepEngine(vae, repr_nn, svgp, b_max=max(int(b_hint), 1), seg_len=1, clip_qs=clipping_qs)
    return _attach(eng, svgp, vae, repr_nn)


def _segment_mean_repeat(cv, segment_ids, repeats):
    """tf.segment_mean + tf.repeat of aux_data_SVGPVAE_sprites (SVGPVAE_model.py:1103-1109); O(n L_character) glue."""
    seg = torch.as_tensor(np.asarray(segment_ids), device=cv.device, dtype=torch.long)
    n_seg = int(seg.max().item()) + 1
    sums = torch.zeros(n_seg, cv.shape[1], dtype=cv.dtype, device=cv.device).index_add_(0, seg, cv)
    cnt = torch.bincount(seg, minlength=n_seg).to(cv.dtype)[:, None]
    return torch.repeat_interleave(sums / cnt, torch.as_tensor(np.asarray(repeats), device=cv.device), dim=0)


def aux_data_SVGPVAE_sprites(data_batch, repr_nn, segment_ids, repeats, engine=None):
    """SVGPVAE_model.py:1086-1115 for arbitrary (segment_ids, repeats): representation network on the frames,
    segment mean per character, repeat, prepend the action ids of the OUTPUT rows.  `engine` defaults to the one
    attached to `repr_nn` (the reference's four-argument call)."""
    images, action_IDs = data_batch
    engine = _engine_of(None, None, repr_nn, engine, images.shape[0], False)
    cv = engine.character_vectors(images.to(engine.dev, _F64))
    cv = _segment_mean_repeat(cv, segment_ids, repeats)
    return torch.cat([action_IDs.to(engine.dev, _F64)[:, None], cv], dim=1).contiguous()


def batching_encode_SVGPVAE(data_batch, vae, clipping_qs=False, repr_nn=None, segment_ids=None, repeats=None, svgp=None,
                            engine=None):
    """SVGPVAE_model.py:939-968 (SPRITES form): (qnet_mu, qnet_var, aux_data) of a batch of train frames."""
    images, action_IDs = data_batch
    eng = _engine_of(svgp, vae, repr_nn, engine, images.shape[0], clipping_qs)
    assert eng.clip_qs == bool(clipping_qs), "engine was built with a different clipping_qs"
    mu, var = eng.encode(images.to(eng.dev, _F64))
    aux = aux_data_SVGPVAE_sprites(data_batch, repr_nn, segment_ids, repeats, eng)
    return mu, var, aux


def precompute_GP_params_SVGPVAE(means, vars, aux_data, svgp, engine=None):
    """SVGPVAE_model.py:989-1023 over ALL train frames: (mean_terms (L, m), inverse Sigma_l (L, m, m)).
    K_nm and the statistics K_mn diag(1/var_l) K_nm, K_mn (mean_l / var_l) run as the float32 streaming kernels
    (float32 is the reference's SPRITES dtype); Sigma_l = K_mm + S_l is inverted in float64, no jitter (:1014)."""
    from . import stream_stats as SS
    eng = engine or _attached(svgp)
    if eng is None:
        raise _lib.SvgpError("precompute_GP_params_SVGPVAE: no step engine attached to `svgp` (pass engine=)")
    dev = eng.dev
    se = eng.params["se"].detach().cpu().tolist()
    kd = SS.kernel_desc(SS.SE_SE if svgp.K_SE else SS.LINEAR_LINEAR, svgp.L_action, svgp.L_character,
                        normalize=svgp.K_obj_normalize and not svgp.K_SE, n_table=eng.n_act, params=se if svgp.K_SE else ())
    f32 = lambda t: t.to(dev, torch.float32).contiguous()
    mt, inv = SS.precompute_GP_params_f32(kd, f32(means), f32(vars), f32(aux_data), f32(eng.params["inducing_index_points"]),
                                          table=f32(eng.params["GPLVM_action"]))
    return mt.double(), inv.double()


def approximate_posterior_params_precomputed_GP_posterior_params(svgp, index_points, mean_terms, sigma_terms, K_mm_inv=None,
                                                                 engine=None):
    """spritesSVGP.approximate_posterior_params_precomputed_GP_posterior_params (SVGPVAE_model.py:610-635) for all L
    channels at once: mean (b, L) = K_bm mean_term_l, B (b, L) = K_bb + diag(-K_bm K_mm^-1 K_mb + K_bm Sigma_l^-1 K_mb)."""
    eng = engine or svgp._engine
    K, Kb, kbb = eng.kernel_matrices(index_points)
    m, b, L = eng.m, Kb.shape[0], mean_terms.shape[0]
    dev = eng.dev
    if K_mm_inv is None:   # :620-622 (with jitter)
        Kj = (K + svgp.jitter * torch.eye(m, dtype=_F64, device=dev))[None].contiguous()
        w = torch.empty(max(int(eng.lib.svgp_spd_inverse_workspace_elems(m, 1)), 1), dtype=_F64, device=dev)
        ld = torch.empty(1, dtype=_F64, device=dev)
        call("svgp_spd_inverse_batched", m, 1, Kj.data_ptr(), ld.data_ptr(), w.data_ptr(),
             torch.cuda.current_stream(dev).cuda_stream)
        K_mm_inv = Kj[0]
    s = torch.cuda.current_stream(dev).cuda_stream
    mean = torch.empty(b, L, dtype=_F64, device=dev)
    mt = mean_terms.to(dev, _F64).contiguous()
    call("svgp_dgemm_batched", 0, 1, b, L, m, 1.0, Kb.data_ptr(), m, 0, mt.data_ptr(), m, 0, 0.0, mean.data_ptr(), L, 0, 1, s)
    # X_l = Sigma_l^-1 - K_mm^-1 ;  B_l = k_bb + rowsum((K_bm X_l) o K_bm)
    X = (sigma_terms.to(dev, _F64) - K_mm_inv.to(dev, _F64)[None]).contiguous()
    P = torch.empty(L, b, m, dtype=_F64, device=dev)
    call("svgp_dgemm_batched", 0, 0, b, m, m, 1.0, Kb.data_ptr(), m, 0, X.data_ptr(), m, m * m, 0.0, P.data_ptr(), m, b * m,
         L, s)
    B = kbb[:, None] + (P * Kb[None]).sum(-1).t()
    return mean, B


def predict_SVGPVAE_sprites_test_character(data_batch, vae, svgp, repr_NN, mean_terms, var_terms, N_context, N_actions,
                                           batch_size_test, segment_ids, repeats, K_mm_inv, context_full_actions=True,
                                           epsilon=None, engine=None, context_draw=None):
    """SVGPVAE_model.py:1118-1195: context / target split of a test-character batch, aux data of the targets from the
    context frames, latents from the precomputed GP posterior (clip [1e-4, 100]), decode, summed squared error / pixels.
    Returns (recon_images_test, target images, recon_loss).  `epsilon` (n_target, L) makes the N(0,1) draw an input.
    context_full_actions=False (:1149-1151): the context of every character is N_context of its N_actions frames chosen
    without replacement -- `context_draw` (n_characters, N_context) integer offsets makes that draw an input too; None draws
    with np.random.choice exactly as the reference does."""
    images, aux_data_target = data_batch
    eng = _engine_of(svgp, vae, repr_NN, engine, batch_size_test, False)
    dev = eng.dev
    n_char = int(batch_size_test / N_actions)
    if context_full_actions:
        context = np.sort(np.array([list(range(i * N_actions, i * N_actions + N_context)) for i in range(n_char)]).reshape(-1))
    else:
        if context_draw is None:
            context_draw = [np.random.choice(range(N_actions), N_context, replace=False) for _ in range(n_char)]
        context_draw = np.asarray(context_draw)
        assert context_draw.shape == (n_char, N_context) and all(len(set(r.tolist())) == N_context for r in context_draw)
        assert context_draw.min() >= 0 and context_draw.max() < N_actions
        context = np.sort(np.array([list(i * N_actions + context_draw[i]) for i in range(n_char)]).reshape(-1))
    target = np.array([x for x in range(batch_size_test) if x not in set(context.tolist())])
    images = images.to(dev, _F64)
    ids = aux_data_target.to(dev, _F64)
    images_context, images_t = images[context].contiguous(), images[target].contiguous()
    aux_t = aux_data_SVGPVAE_sprites((images_context, ids[target]), repr_NN, segment_ids, repeats, eng)
    p_m, p_v = approximate_posterior_params_precomputed_GP_posterior_params(svgp, aux_t, mean_terms, var_terms, K_mm_inv,
                                                                           engine=eng)
    p_v = torch.clamp(p_v, 1e-4, 100.0)
    if epsilon is None:
        epsilon = torch.randn(p_m.shape, dtype=_F64, device=dev)
    z = (p_m + epsilon.to(dev, _F64) * torch.sqrt(p_v)).contiguous()
    recon = eng.decode(z)
    recon_loss = torch.sum((images_t - recon) ** 2) / float(64 * 64 * 3)
    return recon, images_t, recon_loss


# ---------------------------------------------------------------------------------------------
# pre-training of the representation network (SPRITES_experiment.py:139-151,325-357; SPRITES_utils.py:335-368)
# ---------------------------------------------------------------------------------------------
class repr_NN_classification_layer:
    """The `tf.keras.layers.Dense(1000)` of the pre-training phase (SPRITES_experiment.py:139-141): glorot-uniform
    kernel (L_character, n_classes), zero bias; float64 device tensors `W`, `b`."""

    def __init__(self, n_in, n_classes=1000, seed=0, device="cuda:0"):
        lim = math.sqrt(6.0 / (n_in + n_classes))
        rs = np.random.RandomState(seed)
        self.W = torch.tensor(rs.uniform(-lim, lim, (n_in, n_classes)), dtype=_F64, device=device)
        self.b = torch.zeros(n_classes, dtype=_F64, device=device)
        self.reset_metrics()

    def reset_metrics(self):
        """`sess.run(tf.local_variables_initializer())` (SPRITES_experiment.py:304,326): zeroes the total / count pair behind
        the streaming accuracy of forward_pass_pretraining_repr_NN(test_pipeline=True)."""
        self.acc_total, self.acc_count = 0.0, 0.0


def forward_pass_pretraining_repr_NN(frames, labels, repr_NN, classification_layer, test_pipeline=False, engine=None):
    """SPRITES_utils.py:335-368: loss = mean sparse softmax cross-entropy of Dense(repr_nn(frames)) against the character
    ids; with `test_pipeline` also the accuracy of the arg-max prediction -- as in the reference the STREAMING accuracy
    (`_, acc = tf.compat.v1.metrics.accuracy(...)`, :364: the update op's value, i.e. correct / seen over every batch evaluated
    since the metric's local variables were initialised; the pair lives on `classification_layer`, `reset_metrics()` is the
    `tf.local_variables_initializer()` of SPRITES_experiment.py:326).  The in-batch shuffle of :346-351 permutes the
    rows of a mean and is not reproduced.  Runs the representation network, the average pool, the dense product and
    `svgp_softmax_xent` of the library on the engine attached to `repr_NN` (or `engine=`); the training loop around it,
    with its reverse pass and TF1 Adam, is `pretrain_repr_NN`."""
    eng = _engine_of(None, None, repr_NN, engine, frames.shape[0], False)
    dev, Lc, n = eng.dev, eng.Lc, frames.shape[0]
    W, bC = classification_layer.W.to(dev, _F64), classification_layer.b.to(dev, _F64)
    n_classes = W.shape[1]
    lab_all = torch.as_tensor(labels).to(dev, _F64).contiguous()
    emb = eng.character_vectors(frames.to(dev, _F64))                    # chunks of at most b_max frames
    logits = torch.empty(n, n_classes, dtype=_F64, device=dev)
    rowloss, loss, dlog = torch.empty(n, dtype=_F64, device=dev), torch.empty(1, dtype=_F64, device=dev), torch.empty_like(logits)
    eng.stream.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(eng.stream):
        s = eng.stream.cuda_stream
        eng._gemm(0, 0, n, n_classes, Lc, 1.0, emb, Lc, W, n_classes, 0.0, logits, n_classes)
        call("svgp_bias_add", n, n_classes, bC.data_ptr(), logits.data_ptr(), s)
        call("svgp_softmax_xent", n, n_classes, logits.data_ptr(), lab_all.data_ptr(), rowloss.data_ptr(), loss.data_ptr(),
             dlog.data_ptr(), s)
    eng.stream.synchronize()
    if not test_pipeline:
        return loss[0]
    hits = float((torch.argmax(logits, dim=1) == lab_all.long()).sum())
    if not hasattr(classification_layer, "acc_total"):
        classification_layer.acc_total, classification_layer.acc_count = 0.0, 0.0
    classification_layer.acc_total += hits
    classification_layer.acc_count += float(n)
    acc = torch.tensor(classification_layer.acc_total / classification_layer.acc_count, dtype=_F64, device=dev)
    return loss[0], acc


def pretrain_repr_NN(engine, frames, char_IDs, *, nr_epochs, lr, batch_size, n_classes=1000, seed=0, log=print,
                     carry_slots=True):
    """Character classification: embeddings = repr_nn(frames) -> Dense(n_classes) -> mean sparse softmax cross-entropy,
    TF1 Adam on the representation network + the classification layer (which is discarded afterwards).  Updates the
    engine's repr_c* parameters in place; returns the list of (epoch, mean loss, accuracy).

    The reference runs pre-training and the joint phase on ONE tf.train.AdamOptimizer (SPRITES_experiment.py:210-238):
    its beta1/beta2 power accumulators keep advancing through pre-training, so the joint phase starts at global step
    K_pretrain + 1, and with `yes_joint` the m / v slots of the representation network carry over.  Both are reproduced:
    the engine's Adam step counter is advanced by the number of pre-training updates, and (carry_slots, = not
    `yes_fixed`) the repr_* slices of the moments are copied into the engine's."""
    dev, Lc, s = engine.dev, engine.Lc, engine.stream.cuda_stream
    f64 = dict(dtype=_F64, device=dev)
    f32, sfx = engine.f32, engine.sfx
    nd = dict(dtype=engine.ndt, device=dev)
    names = [k for k in engine.shapes if k.startswith("repr_")]
    sizes = [int(np.prod(engine.shapes[k])) for k in names]
    n_rep = sum(sizes)
    theta = torch.zeros(n_rep + Lc * n_classes + n_classes, **f64)
    grad, am, av = torch.zeros_like(theta), torch.zeros_like(theta), torch.zeros_like(theta)
    # float32 networks (the reference's SPRITES dtype, VAE_utils.py:277): the representation network runs on the float32
    # convolution entry points from a float32 copy of its parameters, refreshed from the float64 master vector before every
    # step (as the joint step does); its float32 gradients are cast into the float64 gradient vector; the dense layer, the
    # cross-entropy and TF1 Adam stay float64
    theta_n = torch.zeros(n_rep, **nd) if f32 else theta
    grad_n = torch.zeros(n_rep, **nd) if f32 else grad
    views, gviews, off = {}, {}, 0
    for k, n in zip(names, sizes):
        views[k] = theta_n[off:off + n].view(engine.shapes[k]); gviews[k] = grad_n[off:off + n].view(engine.shapes[k])
        off += n
    master = {k: theta[o:o + n].view(engine.shapes[k]) for k, n, o in zip(names, sizes, np.cumsum([0] + sizes[:-1]))}
    W, gW = theta[off:off + Lc * n_classes].view(Lc, n_classes), grad[off:off + Lc * n_classes].view(Lc, n_classes)
    bC, gb = theta[off + Lc * n_classes:], grad[off + Lc * n_classes:]
    rs = np.random.RandomState(seed)
    lim = math.sqrt(6.0 / (Lc + n_classes))                          # Keras Dense default: glorot_uniform, zero bias
    state = torch.zeros(STATE_LEN, **f64)
    engine.stream.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(engine.stream):
        for k in names:
            master[k].copy_(engine.params[k])
        W.copy_(torch.tensor(rs.uniform(-lim, lim, (Lc, n_classes)), **f64))
        state[STATE["LR"]] = lr
    n = frames.shape[0]
    ones = torch.ones(batch_size, 1, **f64)
    history = []
    for epoch in range(nr_epochs):
        tot, correct, seen = 0.0, 0, 0
        for lo in range(0, n - batch_size + 1, batch_size):
            b = batch_size
            with torch.cuda.stream(engine.stream):
                if f32:
                    call("svgp_cast_f64_f32", n_rep, theta.data_ptr(), theta_n.data_ptr(), s)
                x0 = frames[lo:lo + b].to(engine.ndt).contiguous()
                lab = char_IDs[lo:lo + b].to(_F64).contiguous()
                r, x = [], x0
                for i, lay in enumerate(engine.rep, 1):
                    out = torch.empty(b, lay.Ho, lay.Ho, lay.Co, **nd)
                    lay.forward(x, views[f"repr_c{i}_w"], views[f"repr_c{i}_b"], out, s)
                    r.append(out); x = out
                emb_n = torch.empty(b, Lc, **nd)
                call("svgp_avgpool_fwd" + sfx, b, 64, Lc, x.data_ptr(), emb_n.data_ptr(), s)
                emb = emb_n
                if f32:
                    emb = torch.empty(b, Lc, **f64)
                    call("svgp_cast_f32_f64", b * Lc, emb_n.data_ptr(), emb.data_ptr(), s)
                logits = torch.empty(b, n_classes, **f64)
                engine._gemm(0, 0, b, n_classes, Lc, 1.0, emb, Lc, W, n_classes, 0.0, logits, n_classes)
                call("svgp_bias_add", b, n_classes, bC.data_ptr(), logits.data_ptr(), s)
                rowloss, loss, dlog = torch.empty(b, **f64), torch.empty(1, **f64), torch.empty(b, n_classes, **f64)
                call("svgp_softmax_xent", b, n_classes, logits.data_ptr(), lab.data_ptr(), rowloss.data_ptr(), loss.data_ptr(),
                     dlog.data_ptr(), s)
                engine._gemm(1, 0, Lc, n_classes, b, 1.0, emb, Lc, dlog, n_classes, 0.0, gW, n_classes)      # emb^T dlogits
                engine._gemm(1, 0, 1, n_classes, b, 1.0, ones, 1, dlog, n_classes, 0.0, gb, n_classes)       # column sums
                demb = torch.empty(b, Lc, **f64)
                engine._gemm(0, 1, b, Lc, n_classes, 1.0, dlog, n_classes, W, n_classes, 0.0, demb, Lc)      # dlogits W^T
                demb_n = demb
                if f32:
                    demb_n = torch.empty(b, Lc, **nd)
                    call("svgp_cast_f64_f32", b * Lc, demb.data_ptr(), demb_n.data_ptr(), s)
                dx = torch.empty(b, 8, 8, Lc, **nd)
                call("svgp_avgpool_bwd" + sfx, b, 64, Lc, demb_n.data_ptr(), dx.data_ptr(), s)
                for i in range(3, 0, -1):
                    xin = r[i - 2] if i > 1 else x0
                    dx = engine.rep[i - 1].backward(xin, views[f"repr_c{i}_w"], r[i - 1], dx, gviews[f"repr_c{i}_w"],
                                                    gviews[f"repr_c{i}_b"], engine.scratch, s, need_dx=i > 1, nwg=engine.nwg)
                if f32:
                    call("svgp_cast_f32_f64", n_rep, grad_n.data_ptr(), grad.data_ptr(), s)
                call("svgp_adam_tf1_step", theta.numel(), theta.data_ptr(), grad.data_ptr(), am.data_ptr(), av.data_ptr(),
                     state.data_ptr(), 0.9, 0.999, 1e-8, s)
                state[STATE["ADAM_T"]] += 1.0
                pred = torch.argmax(logits, dim=1)
            engine.stream.synchronize()
            tot += float(loss); seen += b
            correct += int((pred == lab.long()).sum())
        nb = max(1, (n - batch_size) // batch_size + 1)
        history.append((epoch, tot / nb, correct / max(seen, 1)))
        if log is not None and ((epoch + 1) % 50 == 0 or epoch == nr_epochs - 1):
            log(f"repr NN pretraining epoch {epoch}: mean loss {history[-1][1]:.4f}  accuracy {history[-1][2]:.4f}")
    with torch.cuda.stream(engine.stream):
        off_p = 0
        for k, n in zip(names, sizes):
            engine.params[k].copy_(master[k])
            if carry_slots:
                off_e = 0
                for ke, se in engine.shapes.items():
                    if ke == k:
                        break
                    off_e += int(np.prod(se))
                engine.adam_m[off_e:off_e + n].copy_(am[off_p:off_p + n])
                engine.adam_v[off_e:off_e + n].copy_(av[off_p:off_p + n])
            off_p += n
        engine.state[STATE["ADAM_T"]] += state[STATE["ADAM_T"]]
    engine.stream.synchronize()
    return history
