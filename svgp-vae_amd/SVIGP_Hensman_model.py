"""Deep SVIGP_Hensman baseline on the HIP library.

Reference call surface mirrored here (eager float64 CUDA tensors instead of TF graph tensors):
  SVIGP_Hensman(fixed_inducing_points, initial_inducing_points, name, jitter, N_train, dtype, L, fixed_gp_params,
                object_vectors_init, K_obj_normalize=False)                       SVIGP_Hensman_model.py:14-77
  SVIGP_Hensman_decoder(L=16)                                                      VAE_utils.py:394-431
  forward_pass_deep_SVIGP_Hensman(data_batch, vae, svgp) -> 8-tuple                SVIGP_Hensman_model.py:230-289
  predict_deep_SVIGP_Hensman(test_data_batch, vae, svgp)                           SVIGP_Hensman_model.py:292-339
and `SvigpStepEngine.step` = `sess.run([optim_step, ...])` of MNIST_experiment.py:633-640, 700-706.

Kernel matrices + VJP and the decoder + reverse are the rotated-MNIST kernels (same kernel, same decoder
architecture); the variational block is svigp.hip (batched MFMA GEMMs + glue kernels); TF1 Adam on the two flat
vectors theta (decoder, inducing points, kernel hyper-parameters, object vectors) and phi (loc, scale, noise).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import STATE, call
from .engine import MnistStepEngine

_F64 = torch.float64


class SVIGP_Hensman_decoder:
    """VAE_utils.py:394-431: the decoder half of mnistVAE (Dense 128 -> (4,4,8) -> 3 x [up, conv])."""
    dtype = torch.float64

    def __init__(self, L=16, seed=0):
        self.L, self.seed = L, seed


class SVIGP_Hensman:
    """SVIGP_Hensman_model.py:14-77.  Holds the initial values; the engine owns the live parameters."""

    def __init__(self, fixed_inducing_points, initial_inducing_points, name, jitter, N_train, dtype, L, fixed_gp_params,
                 object_vectors_init, K_obj_normalize=False):
        self.fixed_inducing_points, self.fixed_gp_params = bool(fixed_inducing_points), bool(fixed_gp_params)
        self.jitter, self.N_train, self.L, self.K_obj_normalize = float(jitter), float(N_train), int(L), bool(K_obj_normalize)
        self.inducing_index_points = np.asarray(initial_inducing_points, dtype=np.float64)
        self.nr_inducing = len(self.inducing_index_points)
        self.object_vectors = None if object_vectors_init is None else np.asarray(object_vectors_init, dtype=np.float64)
        self.l_GP, self.amplitude, self.noise = 1.0, 1.0, 0.1                     # :49-55, :73
        self._engine = None

    def variable_summary(self):
        e = self._engine
        if e is None:
            return self.l_GP, self.amplitude, self.object_vectors, self.inducing_index_points
        p = e.mn.params
        return (p["l_GP"].clone(), p["amplitude"].clone(), p["object_vectors"].clone() if "object_vectors" in p else None,
                p["inducing_index_points"].clone())


class SvigpStepEngine:
    """Buffers + kernel schedule of the deep SVIGP_Hensman step; wraps a MnistStepEngine for the shared pieces."""

    def __init__(self, vae, svgp, *, b_max=256, lr=1e-3, device="cuda:0", params=None):
        m, M = svgp.nr_inducing, svgp.inducing_index_points.shape[1] - 2
        n_obj = 0 if svgp.object_vectors is None else svgp.object_vectors.shape[0]
        self.mn = MnistStepEngine(m, vae.L, M, n_obj, N_train=svgp.N_train, jitter=svgp.jitter, clip_qs=False, geco=False,
                                  K_obj_normalize=svgp.K_obj_normalize, beta=1.0, lr=lr,
                                  train_ip=not svgp.fixed_inducing_points, train_gp=not svgp.fixed_gp_params,
                                  train_ov=n_obj > 0, b_max=b_max, device=device)
        mn = self.mn
        self.m, self.L, self.b_max, self.svgp, self.dev, self.stream = m, vae.L, b_max, svgp, mn.device, mn.stream
        from .VAE_utils import glorot_uniform_params
        init = dict(glorot_uniform_params(vae.L, vae.seed))
        init.update(inducing_index_points=svgp.inducing_index_points, l_GP=svgp.l_GP, amplitude=svgp.amplitude)
        if n_obj:
            init["object_vectors"] = svgp.object_vectors
        if params:
            init.update({k: v for k, v in params.items() if k in mn.params})
        mn.load_params({k: v for k, v in init.items() if k in mn.params})
        # ---- phi = [loc (L,m) | scale (L,m,m) | noise]  (SVIGP_Hensman_model.py:63-73)
        L = self.L
        f64 = dict(dtype=_F64, device=self.dev)
        n_phi = L * m + L * m * m + 1
        self.phi, self.phi_grad = torch.zeros(n_phi, **f64), torch.zeros(n_phi, **f64)
        self.phi_m, self.phi_v = torch.zeros(n_phi, **f64), torch.zeros(n_phi, **f64)
        cut = lambda t: dict(loc=t[:L * m].view(L, m), scale=t[L * m:L * m + L * m * m].view(L, m, m), noise=t[-1:])
        self.vp, self.vg = cut(self.phi), cut(self.phi_grad)
        self.out = torch.zeros(7, **f64)
        self.sws = torch.zeros(int(mn.lib.svgp_svigp_workspace_elems(b_max, m, L)), **f64)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.stream):
            self.vp["scale"].copy_(torch.eye(m, dtype=_F64).expand(L, m, m))
            self.vp["noise"].fill_(svgp.noise)
            for k in ("loc", "scale", "noise"):
                if params and k in params:
                    self.vp[k].copy_(torch.as_tensor(np.asarray(params[k]) if not torch.is_tensor(params[k]) else params[k],
                                                     dtype=_F64).reshape(self.vp[k].shape))
        self.stream.synchronize()
        svgp._engine = self

    def scalars(self):
        return self.mn.scalars()

    def _fwd(self, images, aux, b):
        mn, s = self.mn, self.stream.cuda_stream
        mn.set_batch_size(b)
        mn.bind(images, aux, None)
        cfg, ws = C.byref(mn.cfg), mn.ws.data_ptr()
        v = lambda n: mn.ws[getattr(mn.wl, n):].data_ptr()
        call("svgp_kernel_matrix_fwd", cfg, mn.theta.data_ptr(), aux.data_ptr(), ws, s)
        call("svgp_svigp_fwd", b, b, self.m, self.L, 784, self.svgp.jitter, v("K"), v("Kn"), v("knn"),
             self.vp["loc"].data_ptr(), self.vp["scale"].data_ptr(), self.vp["noise"].data_ptr(), v("z"),
             self.sws.data_ptr(), s)
        call("svgp_mnist_decoder_fwd", cfg, mn.theta.data_ptr(), images.data_ptr(), ws, s)
        return cfg, ws, v

    def step(self, images, aux, adam=True, backward=True):
        """images (b,28,28,1), aux (b,2+M) float64 CUDA tensors: forward, reverse of -elbo, TF1 Adam, scalar outputs."""
        mn, s, b = self.mn, self.stream.cuda_stream, images.shape[0]
        images, aux = images.contiguous(), aux.contiguous()
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.stream):
            cfg, ws, v = self._fwd(images, aux, b)
            st, N = mn.state.data_ptr(), self.svgp.N_train
            n_part = int(mn.wl.n_part)
            # scalar outputs first: the Adam update below changes the noise they are a function of
            call("svgp_svigp_assemble", b, b, self.m, self.L, 784, N, self.vp["noise"].data_ptr(), v("part_sums"), n_part,
                 self.sws.data_ptr(), self.out.data_ptr(), s)
            if backward:
                call("svgp_mnist_decoder_bwd", cfg, mn.theta.data_ptr(), images.data_ptr(), ws, st, s)
                call("svgp_svigp_bwd", b, b, self.m, self.L, 784, N, v("Kn"), self.vp["loc"].data_ptr(),
                     self.vp["scale"].data_ptr(), self.vp["noise"].data_ptr(), v("zbar"), v("part_sums"), n_part,
                     v("Kbar"), v("Knbar"), v("knnbar"), self.vg["loc"].data_ptr(), self.vg["scale"].data_ptr(),
                     self.vg["noise"].data_ptr(), self.sws.data_ptr(), s)
                call("svgp_kernel_matrix_bwd", cfg, mn.theta.data_ptr(), aux.data_ptr(), ws, s)
                call("svgp_mnist_grad_reduce", cfg, ws, s)
                # decoder weight gradients take the factor n_pix / (2 noise^2) that Zbar took inside svgp_svigp_bwd
                g = mn.ws[mn.wl.grad:mn.wl.grad + mn.pl.n_total]
                f_ptr = self.sws[int(mn.lib.svgp_svigp_scale_offset(b, self.m, self.L)):].data_ptr()
                call("svgp_scale_by_device_scalar", int(mn.pl.n_vae - mn.pl.n_enc), f_ptr, g[mn.pl.n_enc:].data_ptr(), s)
                if adam:
                    call("svgp_adam_tf1_step", mn.pl.n_total, mn.theta.data_ptr(), g.data_ptr(), mn.adam_m.data_ptr(),
                         mn.adam_v.data_ptr(), st, 0.9, 0.999, 1e-8, s)
                    call("svgp_adam_tf1_step", self.phi.numel(), self.phi.data_ptr(), self.phi_grad.data_ptr(),
                         self.phi_m.data_ptr(), self.phi_v.data_ptr(), st, 0.9, 0.999, 1e-8, s)
            call("svgp_ball_finalize", 1, int(bool(adam and backward)), 0, self.out.data_ptr(), st, s)
        return self

    def predict(self, images, aux):
        """predict_deep_SVIGP_Hensman: (recon_images_test, squared error / (28*28))."""
        b = images.shape[0]
        images, aux = images.contiguous(), aux.contiguous()
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.stream):
            self._fwd(images, aux, b)
            rec = self.mn.ws_view("recon", (b, 28, 28, 1)).clone()
            loss = torch.sum((images - rec) ** 2) / 784.0
        self.stream.synchronize()
        return rec, loss

    def outputs(self):
        """forward_pass_deep_SVIGP_Hensman's 8-tuple for the last step."""
        self.stream.synchronize()
        b, o = self.mn.cfg.b, self.out
        return (o[0].clone(), o[1].clone(), o[2].clone(), o[3].clone(), self.mn.ws_view("recon", (b, 28, 28, 1)).clone(),
                o[5].clone(), o[6].clone(), self.mn.ws_view("z", (b, self.L)).clone())

    def grads(self):
        self.stream.synchronize()
        g = dict(self.mn.grads())
        g.update(loc=self.vg["loc"], scale=self.vg["scale"], noise=self.vg["noise"])
        return g


def _engine_of(vae, svgp, b, params=None):
    eng = svgp._engine
    if eng is None or eng.b_max < b:
        eng = SvigpStepEngine(vae, svgp, b_max=max(b, 256), params=params)
    return eng


def forward_pass_deep_SVIGP_Hensman(data_batch, vae, svgp, params=None):
    images, aux = data_batch
    eng = _engine_of(vae, svgp, images.shape[0], params)
    eng.step(images.to(eng.dev, _F64), aux.to(eng.dev, _F64), adam=False, backward=False)
    return eng.outputs()


def predict_deep_SVIGP_Hensman(test_data_batch, vae, svgp):
    images, aux = test_data_batch
    eng = _engine_of(vae, svgp, images.shape[0])
    return eng.predict(images.to(eng.dev, _F64), aux.to(eng.dev, _F64))
