"""Shared pytest configuration: markers, import paths, golden-fixture loader."""
from __future__ import annotations

import os
import sys

import numpy as np
import pytest

TESTS_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(TESTS_DIR)
PKG_DIR = os.path.join(REPO_ROOT, "stereo-matching_amd")
GOLDEN_DIR = os.path.join(TESTS_DIR, "golden")

for p in (PKG_DIR, REPO_ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str):
    """Load a committed golden fixture (plain arrays only, no pickles)."""
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc


def has_gpu() -> bool:
    try:
        import torch

        return bool(torch.cuda.is_available())
    except Exception:
        return False
