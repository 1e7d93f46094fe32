import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Per-test duration budget (the GPU suite runs against a 1 200 s step limit): a test that hangs -- a collective
# waiting for a rank that died, a kernel that never ends -- fails after its budget instead of turning the whole suite into "killed at
# the limit".  pytest-timeout is in the image; without it the budget is simply not enforced.  The slowest GPU test takes ~30 s
# (profiles/r06b_gputests.txt), the slowest CPU test (the host-ASan build) ~4 min on 8 cores.
GPU_TEST_BUDGET_S, CPU_TEST_BUDGET_S = 240, 900


def pytest_collection_modifyitems(config, items):
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(GPU_TEST_BUDGET_S if it.get_closest_marker("gpu") else CPU_TEST_BUDGET_S))


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    g = os.path.join(ROOT, "tests", "golden")
    return (dict(np.load(os.path.join(g, "mnist_cfg2_inputs.npz"))),
            dict(np.load(os.path.join(g, "mnist_cfg2_outputs.npz"))))
