"""Tap-table descriptors for the generic HIP convolution (csrc/conv_taps.hip) and a small layer object that
runs forward / data-gradient / weight-gradient of the Keras layers used by spritesVAE and
sprites_representation_network (VAE_utils.py:294-338, 375-391):

  Conv2D(k=3|2, strides 1|2, padding 'same'|'valid', activation elu|None)           -> ConvLayer(up=False)
  UpSampling2D(2) followed by Conv2D(k=3, strides 1, 'same'|'valid')                 -> ConvLayer(up=True)
      (computed as four parity classes on pre-summed effective weights, 2.25x fewer MACs)

Weights keep the TF layout (kh, kw, cin, cout).  Only descriptor construction and O(#weights) re-layouts
(transpose / effective-weight sums) happen here; every per-pixel operation is a HIP kernel.
"""
import math

import torch

import ctypes as C

from ._lib import ConvDesc, SumJob, call

_F64 = torch.float64


def _T(pi, k):          # tap group of kernel row k for base parity pi (csrc/vae_mnist.hip UpConv3)
    return 1 if k + pi >= 2 else 0


def _desc(n, Hi, Wi, Ci, Ho, Wo, Co, Hs, Ws, sy, sx, osy, osx, ooy, oox, taps, act):
    d = ConvDesc(n=n, Hi=Hi, Wi=Wi, Ci=Ci, Ho=Ho, Wo=Wo, Co=Co, Hs=Hs, Ws=Ws, sy=sy, sx=sx, osy=osy, osx=osx,
                 ooy=ooy, oox=oox, nt=len(taps), act=act)
    assert 1 <= len(taps) <= 16
    for t, (oy, ox, woff) in enumerate(taps):
        d.oy[t], d.ox[t], d.woff[t] = oy, ox, woff
    return d


def same_pad(H, k, s):
    """Keras/TF 'same': output ceil(H/s), total pad max((out-1)*s + k - H, 0), extra on bottom/right."""
    out = math.ceil(H / s)
    tot = max((out - 1) * s + k - H, 0)
    return out, tot // 2


class ConvLayer:
    """One (optionally upsample-fused) convolution layer with bias and optional ELU."""

    def __init__(self, Hi, Ci, Co, k=3, stride=1, padding="same", up=False, elu=True, dtype=_F64):
        self.Hi, self.Ci, self.Co, self.k, self.s, self.up, self.elu = Hi, Ci, Co, k, stride, up, elu
        # float64, or float32 = the reference's dtype for the SPRITES networks (VAE_utils.py:277): the *_f32 entry points
        assert dtype in (torch.float64, torch.float32)
        self.dt, self.sfx = dtype, ("" if dtype == torch.float64 else "_f32")
        if up:
            assert k == 3 and stride == 1
            self.pad = 1 if padding == "same" else 0
            self.Ho = 2 * Hi - 2 + 2 * self.pad
        elif padding == "same":
            self.Ho, self.pad = same_pad(Hi, k, stride)
        else:
            self.Ho, self.pad = (Hi - k) // stride + 1, 0
        self.n_w = k * k * Ci * Co

    # ------------------------------------------------------------------ weight re-layouts (O(#weights)), HIP kernels
    def weights_fwd(self, w, stream=None):
        """Weights in the layout the forward / weight-gradient descriptors index (svgp_upconv_weights for up layers)."""
        if not self.up:
            return w.contiguous()
        we = torch.empty(2, 2, 2, 2, self.Ci, self.Co, dtype=self.dt, device=w.device)
        s = torch.cuda.current_stream(w.device).cuda_stream if stream is None else stream
        call("svgp_upconv_weights" + self.sfx, self.Ci, self.Co, w.contiguous().data_ptr(), we.data_ptr(), s)
        return we

    def weights_bwd(self, w, stream=None, wf=None):
        """Transposed (Co x Ci per tap) weights for the data gradient (svgp_transpose_taps)."""
        wf = self.weights_fwd(w, stream) if wf is None else wf
        nt = wf.numel() // (self.Ci * self.Co)
        wt = torch.empty(wf.shape[:-2] + (self.Co, self.Ci), dtype=self.dt, device=wf.device)
        s = torch.cuda.current_stream(wf.device).cuda_stream if stream is None else stream
        call("svgp_transpose_taps" + self.sfx, nt, self.Ci, self.Co, wf.data_ptr(), wt.data_ptr(), s)
        return wt

    def prepare(self, w, stream, need_bwd=True):
        """Forward-layout and transposed weights for this step, computed once (off the launch chains of the forward and reverse
        passes: the engine issues this for every layer at the start of a step on an idle stream).  forward / backward use them
        when they are handed the same `w`."""
        wf = self.weights_fwd(w, stream)
        self._prep = (w.data_ptr(), wf, self.weights_bwd(w, stream, wf) if need_bwd else None)

    def _cached(self, w, stream, bwd):
        prep = getattr(self, "_prep", None)
        if prep is not None and prep[0] == w.data_ptr() and (not bwd or prep[2] is not None):
            return prep[2] if bwd else prep[1]
        return self.weights_bwd(w, stream) if bwd else self.weights_fwd(w, stream)

    def fold_wgrad(self, gwf, stream=None, out=None):
        """Gradient in forward layout -> gradient of the raw (k,k,Ci,Co) weights (svgp_upconv_fold_wgrad for up layers)."""
        if not self.up:
            return gwf.view(self.k, self.k, self.Ci, self.Co)
        g = torch.empty(3, 3, self.Ci, self.Co, dtype=self.dt, device=gwf.device) if out is None else out
        s = torch.cuda.current_stream(gwf.device).cuda_stream if stream is None else stream
        call("svgp_upconv_fold_wgrad" + self.sfx, self.Ci, self.Co, gwf.data_ptr(), g.data_ptr(), s)
        return g

    @property
    def n_wf(self):
        return 16 * self.Ci * self.Co if self.up else self.n_w

    # ------------------------------------------------------------------ descriptors
    def descs_fwd(self, n, act=None):
        Hi, Ci, Co, Ho, k, s = self.Hi, self.Ci, self.Co, self.Ho, self.k, self.s
        act = (1 if self.elu else 2) if act is None else act
        if not self.up:
            taps = [(ky - self.pad, kx - self.pad, (ky * k + kx) * Ci * Co) for ky in range(k) for kx in range(k)]
            return [_desc(n, Hi, Hi, Ci, Ho, Ho, Co, Ho, Ho, s, s, 1, 1, 0, 0, taps, act)]
        ds = []
        for opy in range(2):            # output row parity
            for opx in range(2):
                by, bx = opy - self.pad, opx - self.pad
                py, px = by & 1, bx & 1
                dY, dX = (by - py) // 2, (bx - px) // 2
                taps = [(dY + ty, dX + tx, ((((py * 2 + px) * 2 + ty) * 2 + tx) * Ci * Co))
                        for ty in range(2) for tx in range(2)]
                ds.append(_desc(n, Hi, Hi, Ci, Ho, Ho, Co, Ho // 2, Ho // 2, 1, 1, 2, 2, opy, opx, taps, act))
        return ds

    def descs_bwd_data(self, n):
        """Descriptors producing d(layer input) (n,Hi,Hi,Ci) from dpre (n,Ho,Ho,Co); weights = weights_bwd."""
        Hi, Ci, Co, Ho, k, s = self.Hi, self.Ci, self.Co, self.Ho, self.k, self.s
        if self.up:
            taps = []
            for py in range(2):
                for ty in range(2):
                    for px in range(2):
                        for tx in range(2):
                            taps.append((py + self.pad - 2 * ty, px + self.pad - 2 * tx,
                                         ((((py * 2 + px) * 2 + ty) * 2 + tx) * Co * Ci)))
            return [_desc(n, Ho, Ho, Co, Hi, Hi, Ci, Hi, Hi, 2, 2, 1, 1, 0, 0, taps, 0)]
        if s == 1:
            taps = [(self.pad - ky, self.pad - kx, (ky * k + kx) * Co * Ci) for ky in range(k) for kx in range(k)]
            return [_desc(n, Ho, Ho, Co, Hi, Hi, Ci, Hi, Hi, 1, 1, 1, 1, 0, 0, taps, 0)]
        assert s == 2 and Hi % 2 == 0
        ds = []
        for py in range(2):
            for px in range(2):
                kys = [ky for ky in range(k) if (py + self.pad - ky) % 2 == 0]
                kxs = [kx for kx in range(k) if (px + self.pad - kx) % 2 == 0]
                taps = [((py + self.pad - ky) // 2, (px + self.pad - kx) // 2, (ky * k + kx) * Co * Ci)
                        for ky in kys for kx in kxs]
                if not taps:       # no kernel row reaches this parity: gradient is zero there
                    taps = [(10 ** 6, 10 ** 6, 0)]
                ds.append(_desc(n, Ho, Ho, Co, Hi, Hi, Ci, Hi // 2, Hi // 2, 1, 1, 2, 2, py, px, taps, 0))
        return ds

    # ------------------------------------------------------------------ execution
    def forward(self, x, w, b, out, stream):
        n = x.shape[0]
        ds = self.descs_fwd(n)
        arr = (ConvDesc * len(ds))(*ds)
        wf = self._cached(w, stream, False)
        assert x.dtype == self.dt and w.dtype == self.dt and out.dtype == self.dt
        call("svgp_conv_taps_fwd" + self.sfx, arr, len(ds), x.data_ptr(), wf.data_ptr(), b.data_ptr(), out.data_ptr(), stream)
        return out

    def backward(self, x, w, out, dout, gw, gb, scratch, stream, need_dx=True, dx=None, nwg=512, deferred=None):
        """dout (n,Ho,Ho,Co) is overwritten with dpre.  gw (k,k,Ci,Co), gb (Co) receive the gradients (gw contiguous: the
        reduction writes it in place).  scratch: buffer (layer dtype) of >= scratch_elems(nwg) elements.  Returns dx
        (n,Hi,Hi,Ci) or None.
        deferred: a DeferredSums -- the two closing reductions (weight / bias partials -> gw, gb) are then NOT launched but
        recorded there (svgp_conv_taps_wgrad_fused_jobs), together with the fold of an up layer's effective-weight gradient,
        which reads the summed gradient; `scratch` must be this layer's own region, untouched until DeferredSums.flush."""
        n = x.shape[0]
        assert x.dtype == self.dt and dout.dtype == self.dt and scratch.dtype == self.dt and gw.dtype == self.dt
        assert gw.is_contiguous() and gb.is_contiguous()
        part_b = scratch[:1024 * 16]
        part_w = scratch[1024 * 16:]
        ds = self.descs_fwd(n, act=0)
        arr = (ConvDesc * len(ds))(*ds)
        gwf = torch.empty(self.n_wf, dtype=self.dt, device=x.device) if self.up else gw
        # dpre = dout * elu'(out) in place, gb = column sums, gwf = weight gradient: one pass (svgp_conv_taps_wgrad_fused)
        if deferred is None:
            call("svgp_conv_taps_wgrad_fused" + self.sfx, arr, len(ds), x.data_ptr(), out.data_ptr() if self.elu else None,
                 dout.data_ptr(), part_w.data_ptr(), part_b.data_ptr(), nwg, self.n_wf, gwf.data_ptr(), gb.data_ptr(), stream)
            if self.up:
                self.fold_wgrad(gwf, stream, out=gw)
        else:
            jobs, n_jobs = (SumJob * 4)(), C.c_int(0)
            call("svgp_conv_taps_wgrad_fused_jobs" + self.sfx, arr, len(ds), x.data_ptr(), out.data_ptr() if self.elu else None,
                 dout.data_ptr(), part_w.data_ptr(), part_b.data_ptr(), nwg, self.n_wf, gwf.data_ptr(), gb.data_ptr(), jobs, 4,
                 C.byref(n_jobs), stream)
            deferred.add(self.dt, [jobs[k] for k in range(n_jobs.value)], keep=(gwf, scratch),
                         fold=(lambda s_, gwf=gwf, gw=gw: self.fold_wgrad(gwf, s_, out=gw)) if self.up else None)
        if not need_dx:
            return None
        db = self.descs_bwd_data(n)
        arrb = (ConvDesc * len(db))(*db)
        wb = self._cached(w, stream, True)
        if dx is None:
            dx = torch.empty(n, self.Hi, self.Hi, self.Ci, dtype=self.dt, device=x.device)
        call("svgp_conv_taps_fwd" + self.sfx, arrb, len(db), dout.data_ptr(), wb.data_ptr(), None, dx.data_ptr(), stream)
        return dx

    def scratch_elems(self, nwg=512):
        return 1024 * 16 + 4 * nwg * self.n_wf


class DeferredSums:
    """The partial-sum reductions of a reverse pass, collected layer by layer and run as ONE launch per element type
    (svgp_sum_partials_multi), followed by the folds of the up layers' effective-weight gradients."""

    def __init__(self):
        self.jobs = {torch.float64: [], torch.float32: []}
        self.folds, self.keep = [], []

    def add(self, dtype, jobs, keep=(), fold=None):
        for j in jobs:                       # copies: the ctypes array they came from does not outlive the layer call
            c = SumJob()
            C.memmove(C.byref(c), C.byref(j), C.sizeof(SumJob))
            self.jobs[dtype].append(c)
        self.keep.extend(keep)               # buffers the recorded jobs read / write stay referenced until the flush
        if fold is not None:
            self.folds.append(fold)

    def flush(self, stream):
        for dt, sfx in ((torch.float64, ""), (torch.float32, "_f32")):
            js = self.jobs[dt]
            if js:
                call("svgp_sum_partials_multi" + sfx, (SumJob * len(js))(*js), len(js), stream)
        for f in self.folds:
            f(stream)
        self.__init__()
