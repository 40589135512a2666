import os
import sys

import pytest

# The oracle is thousands of tiny torch-CPU ops per frame: with one intra-op thread per logical core (256 on the GPU box) every one of
# them pays a thread-pool round trip and the oracle-bound tests run 10x slower (58 s -> 5 s for two of them, measured).  Cap it.
os.environ.setdefault("OMP_NUM_THREADS", "8")
os.environ.setdefault("MKL_NUM_THREADS", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch
        torch.set_num_threads(min(8, os.cpu_count() or 1))
    except Exception:          # noqa: BLE001 - torch is optional for the pure-host tests
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
