"""Per-phase hipGraphs (the data-parallel replay form) must equal the eager single-stream step."""
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_phase_graphs_equal_eager(golden):
    params, images, aux, eps = H.golden_problem(golden)
    a = H.engine_for(params, 256, geco=True)
    b_ = H.engine_for(params, 256, geco=True)
    dev = a.device
    di, da, de = images.to(dev), aux.to(dev), eps.to(dev)
    a.bind(di, da, de); b_.bind(di, da, de)
    for _ in range(3):
        a.run(adam=True)
    a.synchronize()
    b_.capture_phases("step", adam=True)
    for _ in range(3):
        b_.run_phase_graphs("step")
    b_.synchronize()
    assert torch.equal(a.theta, b_.theta)
    assert a.scalars() == b_.scalars()
