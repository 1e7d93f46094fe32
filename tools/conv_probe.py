"""Development probe: the generic tap-table convolution (conv_taps.hip) on the spritesVAE layer shapes at 500 frames:
forward, data gradient and weight gradient, HIP-event timings -> algorithmic TFLOP/s (2 * k*k * Ci * Co * output pixels)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from svgp_vae_amd.conv import ConvLayer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
DT = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.float64      # conv_probe.py [frames] [f32]
s = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for name, lay in (("enc 64x64 3->16 s1", ConvLayer(64, 3, 16, 3, 1, "same", dtype=DT)), ("enc 64x64 16->16 s2", ConvLayer(64, 16, 16, 3, 2, "same", dtype=DT)),
                  ("enc 32x32 16->16 s1", ConvLayer(32, 16, 16, 3, 1, "same", dtype=DT)), ("dec up 32->64 16->16", ConvLayer(32, 16, 16, 3, 1, "same", up=True, dtype=DT)),
                  ("dec 64x64 16->16 s1", ConvLayer(64, 16, 16, 3, 1, "same", dtype=DT)), ("dec 64x64 16->3 s1", ConvLayer(64, 16, 3, 3, 1, "same", dtype=DT))):
    x = torch.randn(n, lay.Hi, lay.Hi, lay.Ci, dtype=DT, device="cuda")
    w = torch.randn(3, 3, lay.Ci, lay.Co, dtype=DT, device="cuda") * 0.1
    b = torch.zeros(lay.Co, dtype=DT, device="cuda")
    out = torch.empty(n, lay.Ho, lay.Ho, lay.Co, dtype=DT, device="cuda")
    dout = torch.randn_like(out)
    gw, gb = torch.empty_like(w), torch.empty_like(b)
    NWG = int(os.environ.get("CONV_NWG", "512"))
    scratch = torch.zeros(lay.scratch_elems(NWG), dtype=DT, device="cuda")
    flops = 2.0 * 9 * lay.Ci * lay.Co * n * lay.Ho * lay.Ho
    if lay.up:
        flops = 2.0 * 4 * lay.Ci * lay.Co * n * lay.Ho * lay.Ho       # four 2x2 parity classes
    tf = timeit(lambda: lay.forward(x, w, b, out, s))
    d2 = dout.clone()
    tb = timeit(lambda: lay.backward(x, w, out, d2, gw, gb, scratch, s, need_dx=True, nwg=NWG))
    tw = timeit(lambda: lay.backward(x, w, out, d2, gw, gb, scratch, s, need_dx=False, nwg=NWG))
    print(f"{name}: fwd {tf*1e6:.0f} us ({flops/tf/1e12:.1f} TF)  bwd(elu+wgrad+dgrad) {tb*1e6:.0f} us  elu+wgrad {tw*1e6:.0f} us "
          f"({flops/tw/1e12:.1f} TF)  dgrad {max(tb-tw,1e-9)*1e6:.0f} us ({flops/max(tb-tw,1e-9)/1e12:.1f} TF)", flush=True)
