"""``bblackjax/smc/ess.py``: ``ess_solver`` (``:46-89``) -- the increment ``delta`` whose importance weights reach the
target effective sample size -- with the root solver fused on the device (``mfm_smc_delta``).  As written in the
reference the solver weighs with ``exp(-delta * logdensity)`` (``:83``); reproduced as is."""
from . import solver
from .base import _engine_of


def ess_solver(logdensity_fn, particles, target_ess: float, max_delta: float, root_solver=solver.dichotomy):
    if root_solver is not solver.dichotomy:
        raise NotImplementedError("only the dichotomy root solver (solver.py:20-82, the reference default) is built")
    eng = _engine_of(particles)
    ll = logdensity_fn(particles)
    from .base import _sharded
    if _sharded(eng):                      # the effective sample size is over ALL particles: every rank solves on the gathered log-likelihoods
        from ...engine import allgather_cat
        ll = allgather_cat(ll.contiguous())
    return eng.ctx.smc_delta(ll, target_ess, max_delta)
