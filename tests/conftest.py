import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order on the GPU box: the parity files — every comparison of the HIP path with the oracle and the golden fixtures — run FIRST;
# the files that start launchers and rehearse multi-rank forms with all ranks on one GPU (dozens of child processes, each a cold `import torch` on a
# fresh box) run LAST, so that whatever happens in a rehearsal, `pytest -x` has already been through the parity evidence.
LAST = ("test_wire_gpu.py", "test_sharded_gpu.py", "test_bench_gpu.py")


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        name = os.path.basename(str(item.fspath))
        return LAST.index(name) + 1 if name in LAST else 0

    items.sort(key=key)  # stable: the order inside each group stays pytest's own


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
