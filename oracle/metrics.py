"""Sample-quality metrics: kernelised Stein discrepancy (IMQ kernel) and maximum mean discrepancy (RBF kernel).

ORACLE (test infrastructure; see oracle/__init__.py).  Follows ``mcmc_utils.py:28-85`` (``stein_disc``) and
``:88-111`` (``max_mean_disc``), float64, evaluated in row blocks so that the N x N pair matrix is never stored.
"""
import numpy as np


def stein_disc(X, grad_logprob, beta=-0.5, block=512):
    """(U-statistic, V-statistic) of ``mcmc_utils.py:28-85``.  ``grad_logprob`` maps [n, d] -> [n, d]."""
    X = np.asarray(X, dtype=np.float64)
    T, d = X.shape
    G = np.asarray(grad_logprob(X), dtype=np.float64)
    b = -beta                                                       # :54
    tot = 0.0
    for s in range(0, T, block):
        x, g = X[s:s + block], G[s:s + block]
        diff = x[:, None, :] - X[None, :, :]                        # :70
        r2 = (diff ** 2).sum(-1)                                    # :71
        gd = ((g[:, None, :] - G[None, :, :]) * diff).sum(-1)
        gg = g @ G.T
        tot += (-4 * b * (b + 1) * r2 / (1 + r2) ** (b + 2) + 2 * b * (d + gd) / (1 + r2) ** (1 + b) + gg / (1 + r2) ** b).sum()   # :74-78
    diag = (2 * b * d + (G ** 2).sum(-1)).sum()                     # disc(x, x)
    return (tot - diag) / (T * (T - 1)), tot / T ** 2               # :85


def max_mean_disc(X, Y, block=512):
    """``mcmc_utils.py:88-111`` (RBF kernel, sigma2 = 1; both sample sets have m rows)."""
    X, Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)
    m = X.shape[0]

    def ksum(A, B):
        tot = 0.0
        for s in range(0, A.shape[0], block):
            r2 = ((A[s:s + block, None, :] - B[None, :, :]) ** 2).sum(-1)
            tot += np.exp(-0.5 * r2).sum()
        return tot

    disc_x, disc_y, disc_xy = ksum(X, X) - m, ksum(Y, Y) - m, ksum(X, Y)     # :106-108
    m2 = m * m
    return disc_x / (m2 - m) - 2 * disc_xy / m2 + disc_y / (m2 - m)          # :110
