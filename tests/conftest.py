import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def unpack_masks(g, prefix=""):
    """Dropout masks stored by oracle/make_golden.py:pack_masks -> list of bool tensors (NCHW / [N,C])."""
    masks, i = [], 0
    while f"{prefix}noise_drop{i}_bits" in g:
        shape = tuple(int(x) for x in g[f"{prefix}noise_drop{i}_shape"])
        n = int(np.prod(shape))
        bits = np.unpackbits(g[f"{prefix}noise_drop{i}_bits"])[:n].astype(bool).reshape(shape)
        masks.append(torch.from_numpy(bits))
        i += 1
    return masks


@pytest.fixture(scope="session")
def golden():
    return load_golden
