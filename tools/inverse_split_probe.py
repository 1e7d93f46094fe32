"""Development probe: the batched SPD inverse (potrf + potri) of 64 matrices of 800 x 800 as ONE call against TWO half-batch calls
on two streams (each with its own workspace and its own look-ahead branch), HIP events via torch."""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from svgp_vae_amd import _lib
lib = _lib.load_library()
DT = torch.float64
m, batch = int(sys.argv[1]) if len(sys.argv) > 1 else 800, int(sys.argv[2]) if len(sys.argv) > 2 else 64
g = torch.Generator(device="cuda").manual_seed(m)
X = torch.randn(batch, m, m + 8, dtype=DT, device="cuda", generator=g)
A = X @ X.transpose(1, 2) / m + 0.05 * torch.eye(m, dtype=DT, device="cuda")
B = A.clone()
ld = torch.zeros(batch, dtype=DT, device="cuda")
main = torch.cuda.current_stream()
s2 = torch.cuda.Stream()
w = torch.zeros(lib.svgp_spd_inverse_workspace_elems(m, batch), dtype=DT, device="cuda")
def parts(nparts):
    cuts = [batch * i // nparts for i in range(nparts + 1)]
    ws = [torch.zeros(lib.svgp_spd_inverse_workspace_elems(m, cuts[i + 1] - cuts[i]), dtype=DT, device="cuda") for i in range(nparts)]
    return cuts, ws
def one():
    _lib.call("svgp_spd_inverse_batched", m, batch, B.data_ptr(), ld.data_ptr(), w.data_ptr(), main.cuda_stream)
cuts2, ws2 = parts(2)
def two_streams():
    s2.wait_stream(main)
    for i, st in enumerate((main, s2)):
        lo, hi = cuts2[i], cuts2[i + 1]
        _lib.call("svgp_spd_inverse_batched", m, hi - lo, B[lo:].data_ptr(), ld[lo:].data_ptr(), ws2[i].data_ptr(), st.cuda_stream)
    main.wait_stream(s2)
def two_serial():
    for i in range(2):
        lo, hi = cuts2[i], cuts2[i + 1]
        _lib.call("svgp_spd_inverse_batched", m, hi - lo, B[lo:].data_ptr(), ld[lo:].data_ptr(), ws2[i].data_ptr(), main.cuda_stream)
def timed(fn, reps=7):
    B.copy_(A); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        B.copy_(A)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[reps // 2]
out = {"m": m, "batch": batch, "one_call_us": timed(one), "two_halves_two_streams_us": timed(two_streams), "two_halves_one_stream_us": timed(two_serial)}
ref = torch.linalg.inv(A[:2]); B.copy_(A); two_streams(); torch.cuda.synchronize()
out["check"] = float((B[:2] - ref).abs().max())
print(json.dumps({k: (round(v, 2) if isinstance(v, float) and k != "check" else v) for k, v in out.items()}))
