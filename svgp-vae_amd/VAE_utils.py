"""mnistVAE with the reference's call surface (VAE_utils.py:99-162), executed by the HIP library.

`mnistVAE(im_width=28, im_height=28, L=16)`, `.encode(images) -> (means, vars)`,
`.decode(latent_samples) -> recon_images`; tensors are float64 CUDA tensors, NHWC, TF weight
layouts.  Parameters are Keras-initialised (glorot_uniform kernels, zero biases).

The other classes the reference's drivers import from `VAE_utils` resolve here too: `spritesVAE`,
`sprites_representation_network` (VAE_utils.py:275,363; SPRITES_experiment.py:13), `SVIGP_Hensman_decoder` (:394;
MNIST_experiment.py:19).  Their code lives beside the engines that run them (sprites.py, SVIGP_Hensman_model.py).
"""
import math

import numpy as np
import torch

VAE_SHAPES = lambda L: [
    ("enc_c1_w", (3, 3, 1, 8)), ("enc_c1_b", (8,)), ("enc_c2_w", (3, 3, 8, 8)), ("enc_c2_b", (8,)),
    ("enc_c3_w", (3, 3, 8, 8)), ("enc_c3_b", (8,)), ("enc_d_w", (32, 2 * L)), ("enc_d_b", (2 * L,)),
    ("dec_d_w", (L, 128)), ("dec_d_b", (128,)), ("dec_c1_w", (3, 3, 8, 8)), ("dec_c1_b", (8,)),
    ("dec_c2_w", (3, 3, 8, 8)), ("dec_c2_b", (8,)), ("dec_c3_w", (3, 3, 8, 1)), ("dec_c3_b", (1,))]


def glorot_uniform_params(L=16, seed=0):
    """Keras default initialisers for the 16 mnistVAE variables (numpy float64)."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shp in VAE_SHAPES(L):
        if name.endswith("_b"):
            out[name] = np.zeros(shp)
            continue
        rf = shp[0] * shp[1] if len(shp) == 4 else 1
        fan_in, fan_out = (rf * shp[2], rf * shp[3]) if len(shp) == 4 else shp
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        out[name] = rng.uniform(-lim, lim, size=shp)
    return out


class mnistVAE:
    dtype = torch.float64

    def __init__(self, im_width=28, im_height=28, L=16, seed=0, device="cuda:0"):
        if (im_width, im_height) != (28, 28):
            raise NotImplementedError("the HIP encoder/decoder are specialised to 28x28x1 rotated-MNIST images")
        self.L = L
        self.device = device
        self.params = {k: torch.tensor(v, dtype=self.dtype) for k, v in glorot_uniform_params(L, seed).items()}
        self._engine = None     # set when bound to a training runtime (SVGPVAE_model._Runtime)
        self._solo = None

    # -- standalone encode/decode run on a private engine (m=1, no GP); inside a training runtime the
    #    parameters are views into that runtime's flat vector and the same kernels are used.
    def _solo_engine(self, b):
        from .engine import MnistStepEngine
        if self._engine is not None and b <= self._engine.b_max:
            return self._engine
        # stand-alone, or more rows than the training runtime was sized for: a private forward-only engine with the
        # current parameter values (the training engine and its optimiser state stay untouched)
        if self._solo is None or self._solo.b_max < b:
            self._solo = MnistStepEngine(1, self.L, 1, 0, b_max=max(b, 256), device=self.device)
        self._solo.load_params({k: v for k, v in self.params.items()})
        return self._solo

    def encode(self, images):
        import ctypes as C
        from ._lib import call
        eng = self._solo_engine(images.shape[0])
        b = images.shape[0]
        saved = (eng.cfg.b, eng.cfg.b_global, eng.cfg.clip_qs)
        eng.set_batch_size(b)
        eng.cfg.clip_qs = 0
        img = images.to(eng.device, torch.float64).contiguous()
        with torch.cuda.stream(eng.stream):
            # `img` may have been cast / copied / produced on torch's current stream just now
            eng.stream.wait_stream(torch.cuda.current_stream(eng.device))
            call("svgp_mnist_encoder_fwd", C.byref(eng.cfg), eng.theta.data_ptr(), img.data_ptr(), eng.ws.data_ptr(),
                 eng.stream.cuda_stream)
        eng.synchronize()
        out = eng.ws_view("qnet_mu", (b, self.L)).clone(), eng.ws_view("qnet_var_raw", (b, self.L)).clone()
        eng.set_batch_size(saved[0], saved[1])
        eng.cfg.clip_qs = saved[2]
        return out

    def decode(self, latent_samples):
        import ctypes as C
        from ._lib import call
        b = latent_samples.shape[0]
        eng = self._solo_engine(b)
        saved = (eng.cfg.b, eng.cfg.b_global)
        eng.set_batch_size(b)
        eng.ws_view("z", (b, self.L)).copy_(latent_samples.to(eng.device, torch.float64))
        dummy = torch.zeros(b, 28, 28, 1, dtype=torch.float64, device=eng.device)
        with torch.cuda.stream(eng.stream):
            eng.stream.wait_stream(torch.cuda.current_stream(eng.device))
            call("svgp_mnist_decoder_fwd", C.byref(eng.cfg), eng.theta.data_ptr(), dummy.data_ptr(),
                 eng.ws.data_ptr(), eng.stream.cuda_stream)
        eng.synchronize()
        out = eng.ws_view("recon", (b, 28, 28, 1)).clone()
        eng.set_batch_size(*saved)
        return out


from .sprites import spritesVAE, sprites_representation_network  # noqa: E402,F401  (VAE_utils.py:275,363)


def __getattr__(name):
    # SVIGP_Hensman_model imports this module for the Keras initialiser: resolve its decoder class on first use
    if name == "SVIGP_Hensman_decoder":
        from .SVIGP_Hensman_model import SVIGP_Hensman_decoder
        return SVIGP_Hensman_decoder
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
