import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library is built in-tree by ``__graft_entry__.build()``; (re)build it if it is missing or older than
    its sources (hipcc cross-compiles without a GPU), so a fresh checkout can run the suite directly."""
    import __graft_entry__ as entry
    entry.build()
    yield


def record_parity(line: str) -> None:
    """Keep the parity evidence a GPU test computes (observed flip counts, worst errors): printed (``pytest -rA -s``) and appended
    to ``gpurun_out/gpu_parity.txt`` -- which ``gpurun`` brings back from the GPU box; the round's copy is committed as
    ``profiles/rNN_gpu_parity.txt``."""
    print(line)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "gpu_parity.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
