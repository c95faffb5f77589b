"""``bblackjax/smc/solver.py``: the dichotomy root solver (``:20-82``, ``eps = 1e-4``, ``max_iter = 100``) is evaluated
inside ``mfm_smc_delta`` together with the function it solves; this name is the selector ``ess_solver`` checks."""


def dichotomy(fun, _delta0, min_delta, max_delta, eps=1e-4, max_iter=100):
    """Host version for scalar ``fun`` (kept for API completeness; the SMC loop uses the fused device kernel)."""
    f_a, f_b = fun(min_delta), fun(max_delta)
    if f_b > 0:
        return max_delta
    if not f_a > 0:
        return float("nan")
    a, b, i = min_delta, max_delta, 0
    while i < max_iter and f_a - f_b > eps:
        mid = 0.5 * (a + b)
        f_mid = fun(mid)
        if f_mid < 0:
            b, f_b = mid, f_mid
        else:
            a, f_a = mid, f_mid
        i += 1
    return a
