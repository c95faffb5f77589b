"""Freeze the parameters of the 16-mode `gaussian-mixture` target (BASELINE configs[1]).

The reference draws them at start-up from jax.random.PRNGKey(0) (multi_modal.py:39-47): 16 modes U(-12.8, 12.8)^2, per-coordinate
variances exp(0.5 N(0, 1)), weights Dirichlet(4 * 1_16).  jax's gamma sampler (behind `dirichlet`) is third-party arithmetic that
is not restated here, and jax cannot be imported in the build container, so -- as SURVEY.md section 8(c) prescribes -- the
parameters are a committed fixture drawn ONCE with the build's own generator: the threefry conventions of mfm_amd/random.py for
the modes and variances (the same call sequence as the reference) and a numpy gamma draw seeded by the weight key.
`mfm_amd.multi_modal.main` constructs the target by the same recipe; tests/test_golden.py checks that it reproduces this file.

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
