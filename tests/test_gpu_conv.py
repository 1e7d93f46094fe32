"""Generic tap-table MFMA convolution (csrc/conv_taps.hip via svgp_vae_amd.conv.ConvLayer) against the oracle's
Keras-semantics convolutions (F.conv2d on CPU, float64): forward, data gradient, weight and bias gradients."""
import pytest
import torch
import torch.nn.functional as F

from oracle import svgpvae_oracle as O

pytestmark = pytest.mark.gpu
DT = torch.float64

CASES = [
    # (Hi, Ci, Co, k, stride, padding, up, elu)   -- the spritesVAE / repr-net / mnistVAE layer types
    (64, 3, 16, 3, 1, "same", False, True),
    (64, 16, 16, 3, 2, "same", False, True),
    (16, 16, 16, 3, 1, "same", False, True),
    (8, 16, 16, 3, 1, "same", True, True),
    (32, 16, 16, 3, 1, "same", True, True),
    (64, 16, 3, 3, 1, "same", False, True),
    (64, 3, 16, 2, 2, "same", False, True),
    (32, 16, 16, 2, 2, "same", False, True),
    (28, 1, 8, 3, 2, "valid", False, True),
    (8, 8, 8, 3, 1, "valid", True, True),
    (14, 8, 1, 3, 1, "same", True, False),
    (20, 5, 7, 3, 1, "same", False, False),
    # 16 input channels = the direct kernels (k_conv16_*): widths that are not multiples of 16, fewer than 16 output channels,
    # 'valid' padding (tap offsets 0..2), stride 2 on an even size with a 2 x 2 kernel, narrow images (several rows per wave step)
    (20, 16, 16, 3, 1, "same", False, True),
    (37, 16, 5, 3, 1, "valid", False, True),
    (22, 16, 9, 3, 2, "same", False, False),
    (10, 16, 16, 3, 1, "valid", True, True),
    (6, 16, 16, 3, 1, "same", False, True),
    (5, 16, 7, 3, 1, "same", True, False),
    (18, 16, 11, 2, 2, "same", False, True),
    # k_conv16_wgrad_grid (16 -> 16, width a multiple of 16): three 16-pixel segments (interior halo columns are real pixels)
    (48, 16, 16, 3, 1, "same", False, True),
    # k_conv16_thin_fwd (16 -> 3) at widths that are not multiples of its 14-column segments / of 16, 'valid' tap offsets, no activation
    (20, 16, 3, 3, 1, "same", False, True),
    (37, 16, 3, 3, 1, "valid", False, False),
    (32, 16, 3, 3, 1, "same", False, True),
]


def _oracle(x, w, b, k, stride, padding, up, elu):
    h = O._upsample2_nhwc(x) if up else x
    y = O._conv2d_nhwc(h, w, b, stride, padding)
    return F.elu(y) if elu else y


@pytest.mark.parametrize("Hi,Ci,Co,k,stride,padding,up,elu", [c for c in CASES if c[1] == 16 or c[1] == 3])
def test_conv16_direct_kernels_float32(Hi, Ci, Co, k, stride, padding, up, elu):
    """The float32 instances of the 16-channel kernels (v_mfma_f32_16x16x4_f32; v_exp_f32 ELU epilogue in the rolling forward
    kernel) against the float64 oracle: forward 2e-6, gradients 2e-5 relative (float32 accumulation over n x Ho x Ho pixels)."""
    from svgp_vae_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(Hi * 31 + Ci * 7 + Co + k)
    n = 3
    x = torch.randn(n, Hi, Hi, Ci, dtype=DT, generator=g)
    w = torch.randn(k, k, Ci, Co, dtype=DT, generator=g) * 0.3
    b = torch.randn(Co, dtype=DT, generator=g) * 0.1
    xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
    want = _oracle(xr, wr, br, k, stride, padding, up, elu)
    gout = torch.randn(*want.shape, dtype=DT, generator=g)
    gx, gw, gb = torch.autograd.grad((want * gout).sum(), (xr, wr, br))
    F32 = torch.float32
    lay = ConvLayer(Hi, Ci, Co, k=k, stride=stride, padding=padding, up=up, elu=elu, dtype=F32)
    dev = "cuda"
    s = torch.cuda.current_stream().cuda_stream
    dx_, dw_, db_ = x.to(dev, F32), w.to(dev, F32), b.to(dev, F32)
    out = torch.full((n, lay.Ho, lay.Ho, Co), float("nan"), dtype=F32, device=dev)
    lay.forward(dx_, dw_, db_, out, s)
    torch.cuda.synchronize()
    rel = lambda a, c: float((a.cpu().double() - c).abs().max() / (c.abs().max() + 1e-300))
    assert rel(out, want.detach()) < 2e-6
    out64 = want.detach().to(dev, F32).contiguous()          # the reverse pass from the exact forward output
    dout = gout.to(dev, F32).clone()
    ggw = torch.zeros(k, k, Ci, Co, dtype=F32, device=dev)
    ggb = torch.zeros(Co, dtype=F32, device=dev)
    scratch = torch.full((lay.scratch_elems(64),), float('nan'), dtype=F32, device=dev)      # every partial that is summed must have been written
    gdx = lay.backward(dx_, dw_, out64, dout, ggw, ggb, scratch, s, nwg=64)
    torch.cuda.synchronize()
    assert rel(ggb, gb) < 2e-5 and rel(ggw, gw) < 2e-5 and rel(gdx, gx) < 2e-5


@pytest.mark.parametrize("Hi,Ci,Co,k,stride,padding,up,elu", CASES)
def test_conv_layer_forward_and_gradients(Hi, Ci, Co, k, stride, padding, up, elu):
    from svgp_vae_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(Hi * 31 + Ci * 7 + Co + k)
    n = 3
    x = torch.randn(n, Hi, Hi, Ci, dtype=DT, generator=g)
    w = torch.randn(k, k, Ci, Co, dtype=DT, generator=g) * 0.3
    b = torch.randn(Co, dtype=DT, generator=g) * 0.1
    xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
    want = _oracle(xr, wr, br, k, stride, padding, up, elu)
    gout = torch.randn(*want.shape, dtype=DT, generator=g)
    gx, gw, gb = torch.autograd.grad((want * gout).sum(), (xr, wr, br))

    lay = ConvLayer(Hi, Ci, Co, k=k, stride=stride, padding=padding, up=up, elu=elu)
    assert lay.Ho == want.shape[1]
    dev = "cuda"
    s = torch.cuda.current_stream().cuda_stream
    dx_, dw_, db_ = x.to(dev), w.to(dev), b.to(dev)
    out = torch.full((n, lay.Ho, lay.Ho, Co), float("nan"), dtype=DT, device=dev)
    lay.forward(dx_, dw_, db_, out, s)
    torch.cuda.synchronize()
    rel = lambda a, c: float((a.cpu() - c).abs().max() / (c.abs().max() + 1e-300))
    assert rel(out, want.detach()) < 1e-12

    dout = gout.to(dev).clone()
    ggw = torch.zeros(k, k, Ci, Co, dtype=DT, device=dev)
    ggb = torch.zeros(Co, dtype=DT, device=dev)
    scratch = torch.full((lay.scratch_elems(64),), float('nan'), dtype=DT, device=dev)       # every partial that is summed must have been written
    gdx = lay.backward(dx_, dw_, out, dout, ggw, ggb, scratch, s, nwg=64)
    torch.cuda.synchronize()
    assert rel(ggb, gb) < 1e-11
    assert rel(ggw, gw) < 1e-11
    assert rel(gdx, gx) < 1e-11


@pytest.mark.parametrize("rows", [3, 5])
def test_conv16_weight_gradient_odd_row_blocks(rows):
    """k_conv16_wgrad_grid walks a task's rows in pairs (two register sets) and re-reads the current row on the last one: run the
    16 -> 16 layer cases with SVGP_CONV_ROWS = 3 and 5 rows per wave (odd counts; 32 = 10 x 3 + 2 and 6 x 5 + 2 leave a short last
    block).  The row count is read once per process, hence the child interpreter."""
    import os, subprocess, sys
    env = dict(os.environ, SVGP_CONV_ROWS=str(rows))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_conv.py"), "-x", "-q", "-k",
                        "64-16-16-3-2 or 16-16-16-3-1-same-False or 32-16-16-3-1-same-True or 32-16-16-2-2 or 48-16-16-3-1 or 64-3-16"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "no tests ran" not in r.stdout
