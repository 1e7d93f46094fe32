"""Tuning aid: the SPRITES kernel-matrix build and its reverse pass alone on the chip (b = 500, m = 800, cosine-normalised linear x linear)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svgp_vae_amd import _lib
from svgp_vae_amd._lib import SpritesKcfg, call
b, m, La, Lc, n_act = 500, 800, 8, 16, 72
kind = sys.argv[1] if len(sys.argv) > 1 else "cos"
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
f64 = dict(dtype=torch.float64, device=dev)
aux = torch.cat([torch.randint(0, n_act, (b, 1), generator=g).double(), torch.randn(b, Lc, generator=g, dtype=torch.float64)], 1).to(dev)
ip = (torch.randn(m, La + Lc, generator=g, dtype=torch.float64) * 1.5).to(dev)
table = (torch.randn(n_act, La, generator=g, dtype=torch.float64) * 1.5).to(dev)
se = torch.tensor([1.0, 1.0, 1.0, 1.0], **f64)
K, Kn, knn = torch.empty(m, m, **f64), torch.empty(b, m, **f64), torch.empty(b, **f64)
Kbar, Knbar, knnbar = torch.randn(m, m, **f64), torch.randn(b, m, **f64), torch.randn(b, **f64)
d_ip, d_tab, d_char, d_se = torch.empty(m, La + Lc, **f64), torch.empty(n_act, La, **f64), torch.empty(b, Lc, **f64), torch.empty(4, **f64)
kc = SpritesKcfg(b=b, m=m, La=La, Lc=Lc, n_act=n_act, normalize=int(kind == "cos"), k_se=int(kind == "se"), rep_weight=1.0)
scr = torch.empty(int(_lib.load_library().svgp_sprites_kernel_bwd_scratch_elems(C.byref(kc))), **f64)
s = torch.cuda.current_stream().cuda_stream
def t(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
fwd = lambda: call("svgp_sprites_kernel_matrix_fwd", C.byref(kc), aux.data_ptr(), ip.data_ptr(), table.data_ptr(), se.data_ptr(), K.data_ptr(), Kn.data_ptr(), knn.data_ptr(), s)
bwd = lambda: call("svgp_sprites_kernel_matrix_bwd", C.byref(kc), aux.data_ptr(), ip.data_ptr(), table.data_ptr(), se.data_ptr(), Kbar.data_ptr(), Knbar.data_ptr(), knnbar.data_ptr(), d_ip.data_ptr(), d_tab.data_ptr(), d_char.data_ptr(), d_se.data_ptr(), scr.data_ptr(), scr.numel(), s)
print(f"{kind}: kernel_matrix_fwd {t(fwd):.1f} us   kernel_matrix_bwd (cols + rows + scatter) {t(bwd):.1f} us")
