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


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def tap_sample(t):
    """The strided sample tests/golden/make_golden.py::tap_sample stores of an activation tap [B,C,H,W]."""
    cs = 4 if t.shape[1] >= 16 else 2
    s = max(1, t.shape[2] // 8)
    return t[:, ::cs, 1::s, 1::s]


PARITY_LOG = os.path.join(ROOT, "gpurun_out", "parity_report.jsonl")


def record_parity(test, **fields):
    """Append one JSON line per parity measurement to gpurun_out/parity_report.jsonl (the run leaves it behind; a
    copy of the builder's own run is tracked under profiles/), and print it for `pytest -s`."""
    import json
    os.makedirs(os.path.dirname(PARITY_LOG), exist_ok=True)
    rec = {"test": test}
    rec.update(fields)
    with open(PARITY_LOG, "a") as f:
        f.write(json.dumps(rec) + "\n")
    print("PARITY", json.dumps(rec))


@pytest.fixture(scope="session")
def oracle():
    from oracle import vqae_oracle
    return vqae_oracle


@pytest.fixture(scope="session")
def amd():
    import vqae_amd
    return vqae_amd
