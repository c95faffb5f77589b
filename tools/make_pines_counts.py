"""Derive the LGCP bin-count grids from the public Finnish-pines point set.

Input : the reference's data file ``finpines.csv`` (126 points in [0,1]^2; path given on the
        command line, default /root/reference/finpines.csv).
Output: mfm_amd/data/pines_counts.npz with ``counts_<n>`` for n in {4, 8, 16, 32, 40} (flat, row-major),
        binned exactly as ``cox_process_utils.py:29-56`` does (upper-edge points go into the
        last bin).  Only these derived counts are committed, not the point set.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.targets import pines_bin_counts  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/finpines.csv"
pts = np.genfromtxt(src, delimiter=",")
assert pts.shape == (126, 2)
out = {f"counts_{n}": pines_bin_counts(pts, n).reshape(-1).astype(np.int16) for n in (4, 8, 16, 32, 40)}
for k, v in out.items():
    assert v.sum() == 126, k
np.savez_compressed(os.path.join(ROOT, "mfm_amd", "data", "pines_counts.npz"), **out)
print({k: int(v.sum()) for k, v in out.items()})
