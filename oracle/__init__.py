"""CPU oracle for the MFM inner loop -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a float64 numpy restatement of the reference's algorithm for the
hot path named by BASELINE.json (albcab/mfm: ``exe_flow_matching.py:56-449``,
``bblackjax/mcmc/{mala,diffusions,proposal}.py``, ``distributions.py``).  Every
function cites the reference file:line it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the timed CPU port,
never as the thing shipped.  Nothing under ``mfm_amd/`` imports it; the product
path fails loudly when the HIP library is missing.

PARITY UNPINNED.  The reference is pure Python on JAX and cannot be imported in
the build container (no jax / flax / optax / jaxopt / chex wheels, no network),
and it ships no tests, golden vectors or fixtures for this path (SURVEY.md
section 4, section 8c).  The third-party arithmetic it leans on (``jax.random``
threefry conventions, ``jax.experimental.ode.odeint``, ``optax.adamw``,
``jaxopt.Bisection``, flax initialisers; pins in ``environment.yaml``) is
restated from the published algorithms of those packages.  What pins the
restatement instead (tests/test_oracle_*.py):

* Threefry-2x32 against the Random123 known-answer vectors;
* target value / gradient / Hessian-vector closed forms against
  ``torch.autograd`` (float64);
* the vector-field MLP forward / JVP / parameter gradient against
  ``torch.func`` / autograd;
* the AdamW chain against ``torch.optim.AdamW``;
* Dormand-Prince against closed-form linear flows and ``scipy`` RK45;
* the MALA energy algebra against a literal per-chain transcription.

``oracle/cref/`` holds ``libmfm_ref``: the same arithmetic for the headline configuration (PhiFour target, MALA step, relu
network forward / JVP / parameter gradient, adaptive Dopri5 CNF solves with the Hutchinson log-det) in C with OpenMP, one chain
per task -- checked function by function against this package (tests/test_oracle_cref.py) and timed as the CPU baseline of
``bench.py``.  Same standing: test infrastructure, parity unpinned.

Fixtures under ``tests/golden/`` are OUTPUTS OF THIS ORACLE (script:
``tools/make_golden.py``), frozen so the GPU tests have fixed inputs and
expected outputs; they are not outputs of the reference.
"""
