"""Development probe: the convolution entry points (conv_taps.hip) on the spritesVAE layer shapes at 500 frames.
Descriptors, effective / transposed weights and outputs are built once; each C entry point is then timed alone with HIP events
over back-to-back launches -> algorithmic TFLOP/s (2 * taps * Ci * Co * output pixels; up layers: the four 2x2 parity classes).
    python tools/conv_probe.py [frames] [f32]          SVGP_CONV_DIRECT=0: the workgroup-tiled kernels only"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from svgp_vae_amd.conv import ConvLayer
from svgp_vae_amd._lib import ConvDesc, call
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
DT = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.float64
SFX = "_f32" if DT == torch.float32 else ""
s = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
LAYERS = (("enc_c1 64x64 3->16 s1", ConvLayer(64, 3, 16, 3, 1, "same", dtype=DT)), ("enc_c2 64x64 16->16 s2", ConvLayer(64, 16, 16, 3, 2, "same", dtype=DT)),
          ("enc_c3 32x32 16->16 s1", ConvLayer(32, 16, 16, 3, 1, "same", dtype=DT)), ("dec_c5 up 32->64 16->16", ConvLayer(32, 16, 16, 3, 1, "same", up=True, dtype=DT)),
          ("dec_c6 64x64 16->16 s1", ConvLayer(64, 16, 16, 3, 1, "same", dtype=DT)), ("dec_c7 64x64 16->3 s1", ConvLayer(64, 16, 3, 3, 1, "same", dtype=DT)),
          ("repr_c1 64x64 3->16 k2 s2", ConvLayer(64, 3, 16, 2, 2, "same", dtype=DT)), ("enc_c5 16x16 16->16 s1", ConvLayer(16, 16, 16, 3, 1, "same", dtype=DT)))
NWG = int(os.environ.get("CONV_NWG", "512"))
ONLY = os.environ.get("CONV_PROBE_ONLY")
for li, (name, lay) in enumerate(LAYERS):
    if ONLY is not None and li != int(ONLY):
        continue
    x = torch.randn(n, lay.Hi, lay.Hi, lay.Ci, dtype=DT, device="cuda")
    w = torch.randn(lay.k, lay.k, lay.Ci, lay.Co, dtype=DT, device="cuda") * 0.1
    b = torch.zeros(lay.Co, dtype=DT, device="cuda")
    out = torch.empty(n, lay.Ho, lay.Ho, lay.Co, dtype=DT, device="cuda")
    dout = torch.randn_like(out)
    gb = torch.empty_like(b)
    scratch = torch.zeros(lay.scratch_elems(NWG), dtype=DT, device="cuda")
    part_b, part_w = scratch[:1024 * 16], scratch[1024 * 16:]
    taps = 4 if lay.up else lay.k * lay.k
    flops = 2.0 * taps * lay.Ci * lay.Co * n * lay.Ho * lay.Ho
    df = lay.descs_fwd(n); af = (ConvDesc * len(df))(*df)
    dw_ = lay.descs_fwd(n, act=0); aw = (ConvDesc * len(dw_))(*dw_)
    db_ = lay.descs_bwd_data(n); ab = (ConvDesc * len(db_))(*db_)
    wf, wb = lay.weights_fwd(w, s), lay.weights_bwd(w, s)
    gwf = torch.empty(lay.n_wf, dtype=DT, device="cuda")
    dx = torch.empty_like(x)
    tf = timeit(lambda: call("svgp_conv_taps_fwd" + SFX, af, len(df), x.data_ptr(), wf.data_ptr(), b.data_ptr(), out.data_ptr(), s))
    tw = timeit(lambda: call("svgp_conv_taps_wgrad_fused" + SFX, aw, len(dw_), x.data_ptr(), out.data_ptr(), dout.data_ptr(),
                             part_w.data_ptr(), part_b.data_ptr(), NWG, lay.n_wf, gwf.data_ptr(), gb.data_ptr(), s))
    td = timeit(lambda: call("svgp_conv_taps_fwd" + SFX, ab, len(db_), dout.data_ptr(), wb.data_ptr(), None, dx.data_ptr(), s))
    gbytes = x.numel() * x.element_size() + out.numel() * out.element_size()
    print(f"{name}: fwd {tf*1e6:.0f} us ({flops/tf/1e12:.1f} TF, {gbytes/tf/1e12:.2f} TB/s)  elu'+bias+wgrad {tw*1e6:.0f} us ({flops/tw/1e12:.1f} TF)  "
          f"dgrad {td*1e6:.0f} us ({flops/td/1e12:.1f} TF)", flush=True)
