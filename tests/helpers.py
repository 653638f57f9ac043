"""shared helpers for the parity tests"""
import glob
import os

import numpy as np

from conftest import GOLDEN


def golden(pattern):
    files = sorted(glob.glob(os.path.join(GOLDEN, pattern)))
    assert files, f"no golden fixtures match {pattern}"
    return files


def load(path):
    return np.load(path, allow_pickle=False)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(np.max(np.abs(b)), 1e-300)
    return float(np.max(np.abs(a - b)) / denom)
