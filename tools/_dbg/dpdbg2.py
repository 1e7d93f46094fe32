import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import numpy as np, torch
from tests import helpers as H
g = os.path.join(R, "tests", "golden")
golden = (dict(np.load(os.path.join(g, "mnist_cfg2_inputs.npz"))), dict(np.load(os.path.join(g, "mnist_cfg2_outputs.npz"))))
for b in (256, 210):
    for adam in (False, True):
        params, images, aux, eps = H.golden_problem(golden, rows=slice(0, b))
        single = H.engine_for(params, b, geco=True); dev = single.device
        single.bind(images.to(dev), aux.to(dev), eps.to(dev))
        print(b, adam, "before", single.state.cpu().numpy().round(5).tolist())
        single.run(adam=adam); single.synchronize()
        print(b, adam, "after ", single.state.cpu().numpy().round(5).tolist(), flush=True)
