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

from ._lib import ConvDesc, call

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

    # ------------------------------------------------------------------ weight re-layouts (O(#weights))
    def weights_fwd(self, w, stream=None):
        """Weights in the layout the forward / weight-gradient descriptors index (svgp_upconv_weights for up layers)."""
        if not self.up:
            return w.contiguous()
        we = torch.empty(2, 2, 2, 2, self.Ci, self.Co, dtype=self.dt, device=w.device)
        s = torch.cuda.current_stream(w.device).cuda_stream if stream is None else stream
        call("svgp_upconv_weights" + self.sfx, self.Ci, self.Co, w.contiguous().data_ptr(), we.data_ptr(), s)
        return we

    def weights_bwd(self, w, stream=None):
        """Transposed (Co x Ci per tap) weights for the data gradient."""
        return self.weights_fwd(w, stream).transpose(-1, -2).contiguous()

    def fold_wgrad(self, gwf, stream=None):
        """Gradient in forward layout -> gradient of the raw (k,k,Ci,Co) weights (svgp_upconv_fold_wgrad for up layers)."""
        if not self.up:
            return gwf.view(self.k, self.k, self.Ci, self.Co)
        g = torch.empty(3, 3, self.Ci, self.Co, dtype=self.dt, device=gwf.device)
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
        wf = self.weights_fwd(w, stream)
        assert x.dtype == self.dt and w.dtype == self.dt and out.dtype == self.dt
        call("svgp_conv_taps_fwd" + self.sfx, arr, len(ds), x.data_ptr(), wf.data_ptr(), b.data_ptr(), out.data_ptr(), stream)
        return out

    def backward(self, x, w, out, dout, gw, gb, scratch, stream, need_dx=True, dx=None, nwg=512):
        """dout (n,Ho,Ho,Co) is overwritten with dpre.  gw (k,k,Ci,Co), gb (Co) receive the gradients.
        scratch: buffer (layer dtype) of >= scratch_elems(nwg) elements.  Returns dx (n,Hi,Hi,Ci) or None."""
        n = x.shape[0]
        assert x.dtype == self.dt and dout.dtype == self.dt and scratch.dtype == self.dt and gw.dtype == self.dt
        part_b = scratch[:1024 * 16]
        part_w = scratch[1024 * 16:]
        ds = self.descs_fwd(n, act=0)
        arr = (ConvDesc * len(ds))(*ds)
        gwf = torch.empty(self.n_wf, dtype=self.dt, device=x.device)
        # dpre = dout * elu'(out) in place, gb = column sums, gwf = weight gradient: one pass (svgp_conv_taps_wgrad_fused)
        call("svgp_conv_taps_wgrad_fused" + self.sfx, arr, len(ds), x.data_ptr(), out.data_ptr() if self.elu else None,
             dout.data_ptr(), part_w.data_ptr(), part_b.data_ptr(), nwg, self.n_wf, gwf.data_ptr(), gb.data_ptr(), stream)
        gw.copy_(self.fold_wgrad(gwf, stream))
        if not need_dx:
            return None
        db = self.descs_bwd_data(n)
        arrb = (ConvDesc * len(db))(*db)
        wb = self.weights_bwd(w, stream)
        if dx is None:
            dx = torch.empty(n, self.Hi, self.Hi, self.Ci, dtype=self.dt, device=x.device)
        call("svgp_conv_taps_fwd" + self.sfx, arrb, len(db), dout.data_ptr(), wb.data_ptr(), None, dx.data_ptr(), stream)
        return dx

    def scratch_elems(self, nwg=512):
        return 1024 * 16 + 4 * nwg * self.n_wf
