import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test limit (pytest-timeout; ignored when the plugin is absent)")


def pytest_collection_modifyitems(config, items):
    """A hung GPU test must not hold the box until the runner's own limit: every test gets a per-test timeout
    (pytest-timeout, when installed) unless it sets its own."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(420))


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o
