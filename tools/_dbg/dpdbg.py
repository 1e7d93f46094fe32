import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import numpy as np, torch
from tests import helpers as H
from tests.test_gpu_dp_virtual import _lockstep
from svgp_vae_amd.engine import shard_rows
g = os.path.join(R, "tests", "golden")
golden = (dict(np.load(os.path.join(g, "mnist_cfg2_inputs.npz"))), dict(np.load(os.path.join(g, "mnist_cfg2_outputs.npz"))))
for G, b, geco in [(2, 210, True), (3, 210, True), (3, 210, False), (3, 240, True), (2, 140, True), (8, 256, True), (5, 256, True)]:
    params, images, aux, eps = H.golden_problem(golden, rows=slice(0, b))
    single = H.engine_for(params, b, geco=geco); dev = single.device
    di, da, de = images.to(dev), aux.to(dev), eps.to(dev)
    single.bind(di, da, de)
    ranks = []
    for r in range(G):
        lo, hi = shard_rows(b, G, r)
        e = H.engine_for(params, hi - lo, geco=geco, rank=r, world_size=G)
        e.set_batch_size(hi - lo, b)
        e.bind(di[lo:hi].contiguous(), da[lo:hi].contiguous(), de[lo:hi].contiguous())
        ranks.append(e)
    single.run(adam=False); single.synchronize(); _lockstep(ranks, adam=False)
    ref, sc = single.scalars(), ranks[0].scalars()
    print(G, b, geco, {k: (round(sc[k], 6), round(ref[k], 6)) for k in ("elbo", "recon_loss", "kl_term", "ce_term", "inside_elbo")},
          "grad relerr", H.relerr(ranks[0].ws_view("gradC"), single.ws_view("gradC")), flush=True)
