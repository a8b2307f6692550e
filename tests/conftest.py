import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the host program's job (include/ipc_amd.h, "environment"): before the first HIP call of the process
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


def pytest_collection_modifyitems(config, items):
    """The two whole faithful runs (C5: 25 000 candidates, C4: 4 450) go first: they are the longest tests of the GPU suite and
    run a third slower at its end (209 s / 91 s against 156 s / 65 s in a fresh process -- dozens of engines, streams and
    workspaces later), and the suite has a wall-clock limit."""
    first = [it for it in items if it.name in ("test_c5_full_faithful_run", "test_c4_full_faithful_run")]
    if first:
        rest = [it for it in items if it not in first]
        items[:] = first + rest
