import os
import sys
from pathlib import Path

# The CPU tests are many tiny torch ops; on a box whose vCPUs are shared, 8 OpenMP threads spin on
# each other (measured here: 24 s against 1.2 s on one thread for 200 small steps). Two threads
# unless the environment says otherwise (the GPU box sets OMP_NUM_THREADS itself); set before torch
# loads so that the world-size-2 gloo subprocesses inherit it.
os.environ.setdefault("OMP_NUM_THREADS", "2")

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str) -> dict:
    """Load one committed golden fixture (made by tools/make_golden.py from the reference)."""
    with np.load(GOLDEN / f"{name}.npz", allow_pickle=False) as z:
        out = {}
        for k in z.files:
            a = z[k]
            out[k] = torch.from_numpy(a) if a.ndim > 0 else a.item()
        return out


def golden_names(prefix: str) -> list[str]:
    return sorted(p.stem for p in GOLDEN.glob(f"{prefix}*.npz"))


@pytest.fixture(scope="session")
def device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
