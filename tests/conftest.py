import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "modular-trajectory-prediction-toolkit_amd")
for p in (PKG_DIR, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def g1_inputs(g):
    """Rebuild a G1 fixture's inputs (X, C0) from its stored seeds."""
    import datagen
    X = datagen.make_encodings(int(g["seed"]), int(g["n"]), int(g["d"]), int(g["k_true"]),
                               int(g["spread"]), int(g["noise"]))
    C0 = X[g["idx0"]].copy()
    return X, C0


G1_NAMES = ["g1_n2048_d96_k32", "g1_n6000_d96_k200", "g1_n4096_d128_k64", "g1_hard_n5000_d96_k64"]


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    # -m gpu tests are the parity tests proper: no silent skip, no CPU fallback
    assert torch.cuda.is_available(), "tests marked gpu need a visible MI355X"
    return torch.device("cuda:0")
