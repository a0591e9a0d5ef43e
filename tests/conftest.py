import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


def SCHED():
    """The per-thread scheduling options of the batched fit (labelany3d_amd.options.sched): tests pin an engine / launch order /
    kernel build through it (round 4: the library reads the LA3D_* environment once, so per-test switches are per-call options)."""
    from labelany3d_amd.options import sched

    return sched
