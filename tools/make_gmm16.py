"""Freeze the parameters of the 16-mode `gaussian-mixture` target (BASELINE configs[1]).

The reference draws them at start-up from jax.random.PRNGKey(0) (multi_modal.py:39-47): 16 modes U(-12.8, 12.8)^2, per-coordinate
variances exp(0.5 N(0, 1)), weights Dirichlet(4 * 1_16).  jax cannot be imported in the build container, so the draws are made
with the build's own restatement of jax.random (mfm_amd/random.py: threefry conventions; `dirichlet` = jax's log-space
Marsaglia-Tsang gamma sampler on split(key, 16) + softmax, restated from the published source of jax 0.4.26 and checked
distributionally in tests/test_oracle_prng.py -- third-party arithmetic, parity unpinned) in the reference's call sequence, and
frozen here so the GPU tests and bench.py have a fixed target.  `mfm_amd.multi_modal.main` constructs the target by the same
recipe; tests/test_oracle_replay_final.py checks that it reproduces this file.

Usage: python tools/make_gmm16.py  ->  tests/golden/gmm16_params.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mfm_amd.multi_modal import gmm16_parameters  # noqa: E402

if __name__ == "__main__":
    modes, covs, weights = gmm16_parameters()
    path = os.path.join(ROOT, "tests", "golden", "gmm16_params.npz")
    np.savez(path, modes=modes, covs=covs, weights=weights)
    print(path, modes.shape, covs.shape, weights.shape, weights.sum())
