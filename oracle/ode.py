"""Adaptive Dormand-Prince 5(4) as in ``jax.experimental.ode.odeint`` + the CNF transforms.

ORACLE (test infrastructure; see oracle/__init__.py).  ``odeint`` is a third-party
dependency of the reference (jax 0.4.26, ``environment.yaml:101``; call site
``exe_flow_matching.py:13,345-349``) that is absent from ``/root/reference``; this is a
restatement of its published algorithm (PARITY UNPINNED; pinned against closed-form
linear flows and scipy RK45 in tests/test_oracle_ode.py):

* Dopri5 tableau, FSAL, error = 5th - 4th order difference;
* ``err_ratio = sqrt(mean((err / (atol + rtol max(|y0|, |y1|)))^2))`` over ALL state
  components -- here ``d + 1`` (x and the log-det), accept iff ``<= 1``;
* ``dt <- dt * clip(0.9 ratio^(-1/5), dfactor, 10)``, ``dfactor = 1 if ratio < 1 else 0.2``,
  ``dt * 10`` if ``ratio == 0``;
* first step by Hairer-Norsett-Wanner II.4 with order 4 (one extra RHS evaluation);
* ``while t < target and i < mxstep and dt > 0`` (``i`` counts ATTEMPTED steps and restarts
  per output time); the value at an output time is the 4th-order polynomial fitted through
  ``y0, y_mid, y1, f0, f1`` of the last accepted step -- not a step end-point.

Under ``jax.vmap`` every chain runs its own adaptive sequence (finished chains are masked);
the batched loop below does the same with explicit masks.

``transform_and_logdet`` / ``inverse_and_logdet`` follow ``exe_flow_matching.py:206-221``
and ``:223-242``: the Hutchinson probe ``z`` is drawn ONCE per solve (same key at every RHS
call), the exact mode takes ``trace(jacfwd(v))``.
"""
import numpy as np

from . import prng

ALPHA = np.array([1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0])
BETA = [
    np.array([1 / 5]),
    np.array([3 / 40, 9 / 40]),
    np.array([44 / 45, -56 / 15, 32 / 9]),
    np.array([19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729]),
    np.array([9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656]),
    np.array([35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]),
]
C_SOL = np.array([35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0])
C_ERR = np.array([35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
                  -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1.0 / 60.0])
C_MID = np.array([6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2,
                  -2691868925 / 45128329728 / 2, 187940372067 / 1594534317056 / 2,
                  -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2])


def _norm(a):
    return np.sqrt((a * a).sum(1))


def initial_step_size(fun, t0, y0, order, rtol, atol, f0):
    scale = atol + np.abs(y0) * rtol
    d0 = _norm(y0 / scale)
    d1 = _norm(f0 / scale)
    with np.errstate(divide="ignore", invalid="ignore"):
        h0 = np.where((d0 < 1e-5) | (d1 < 1e-5), 1e-6, 0.01 * d0 / d1)
    y1 = y0 + h0[:, None] * f0
    f1 = fun(y1, t0 + h0)
    d2 = _norm((f1 - f0) / scale) / h0
    with np.errstate(divide="ignore", invalid="ignore"):
        h1 = np.where((d1 <= 1e-15) & (d2 <= 1e-15), np.maximum(1e-6, h0 * 1e-3),
                      (0.01 / np.maximum(d1, d2)) ** (1.0 / (order + 1.0)))
    return np.minimum(100.0 * h0, h1)


def runge_kutta_step(fun, y0, f0, t0, dt):
    k = [f0]
    for i in range(6):
        ti = t0 + dt * ALPHA[i]
        yi = y0 + dt[:, None] * sum(b * kk for b, kk in zip(BETA[i], k))
        k.append(fun(yi, ti))
    y1 = dt[:, None] * sum(c * kk for c, kk in zip(C_SOL, k)) + y0
    y1_err = dt[:, None] * sum(c * kk for c, kk in zip(C_ERR, k))
    return y1, k[-1], y1_err, k


def interp_fit_dopri(y0, y1, k, dt):
    dt = dt[:, None]
    y_mid = y0 + dt * sum(c * kk for c, kk in zip(C_MID, k))
    dy0, dy1 = k[0], k[-1]
    a = -2.0 * dt * dy0 + 2.0 * dt * dy1 - 8.0 * y0 - 8.0 * y1 + 16.0 * y_mid
    b = 5.0 * dt * dy0 - 3.0 * dt * dy1 + 18.0 * y0 + 14.0 * y1 - 32.0 * y_mid
    c = -4.0 * dt * dy0 + dt * dy1 - 11.0 * y0 - 5.0 * y1 + 16.0 * y_mid
    return [a, b, c, dt * dy0, y0]


def optimal_step_size(last_step, ratio, safety=0.9, ifactor=10.0, dfactor=0.2, order=5.0):
    dfac = np.where(ratio < 1, 1.0, dfactor)
    with np.errstate(divide="ignore"):
        factor = np.minimum(ifactor, np.maximum(ratio ** (-1.0 / order) * safety, dfac))
    return np.where(ratio == 0, last_step * ifactor, last_step * factor)


def odeint(fun, y0, ts, rtol, atol, mxstep, stats=None, replay=None):
    """Batched: y0 [B, n]; ``fun(y [B, n], t [B]) -> [B, n]``.  Returns ``[len(ts), B, n]``.

    ``stats`` (optional dict) receives the attempted-step counts and, per chain, the whole step sequence:
    ``dt_seq [B, A + 1]`` (``[:, j]`` = step size of attempt j, ``[:, 0]`` the initial step; zero past a chain's
    last attempt), ``acc_seq [B, A]`` (accepted?), ``ratio_seq [B, A]`` (error ratio of attempt j) and
    ``dt_own [B, A + 1]`` (the step sizes the controller itself chose: equal to ``dt_seq`` without ``replay``).

    ``replay = dict(dt=[B, >= A + 1], acc=[B, >= A])`` is PARITY INSTRUMENTATION (no counterpart in the reference):
    the solve takes the prescribed step sizes and accept decisions instead of its controller's (which is still
    evaluated and recorded), so two implementations can be compared stage for stage on the SAME step sequence
    instead of through the chaotic float32-vs-float64 controller decisions (tests/test_gpu_replay.py)."""
    B = y0.shape[0]
    t = np.full(B, float(ts[0]))
    f = fun(y0, t)
    dt = np.clip(initial_step_size(fun, t, y0, 4, rtol, atol, f), 0.0, np.inf)
    rec = stats is not None
    dt_seq, acc_seq, ratio_seq, dt_own, act_seq = [], [], [], [], []
    if rec:
        dt_own.append(dt.copy())
    if replay is not None:
        rp_dt, rp_acc = np.asarray(replay["dt"], dtype=np.float64), np.asarray(replay["acc"]).astype(bool)
        dt = rp_dt[:, 0].copy()
    if rec:
        dt_seq.append(dt.copy())
    y, last_t = y0.copy(), t.copy()
    coeff = [y0.copy() for _ in range(5)]
    outs = [y0]
    n_att = np.zeros(B, dtype=np.int64)
    n_evals = 2
    rows = np.arange(B)
    for target in ts[1:]:
        i = np.zeros(B, dtype=np.int64)
        while True:
            active = (t < target) & (i < mxstep) & (dt > 0)
            if not active.any():
                break
            ny, nf, nerr, k = runge_kutta_step(fun, y, f, t, dt)
            n_evals += 6
            nt = t + dt
            tol = atol + rtol * np.maximum(np.abs(y), np.abs(ny))
            ratio = np.sqrt(((nerr / tol) ** 2).mean(1))
            ncoeff = interp_fit_dopri(y, ny, k, dt)
            ndt = np.clip(optimal_step_size(dt, ratio), 0.0, np.inf)
            acc = active & (ratio <= 1.0)
            if rec:
                act_seq.append(active.copy()); ratio_seq.append(ratio.copy()); dt_own.append(ndt.copy())
            if replay is not None:
                j = np.minimum(n_att, rp_acc.shape[1] - 1)
                acc = active & rp_acc[rows, j]
                ndt = rp_dt[rows, np.minimum(n_att + 1, rp_dt.shape[1] - 1)]
            if rec:
                acc_seq.append(acc.copy())
            m = acc[:, None]
            coeff = [np.where(m, nc, c) for nc, c in zip(ncoeff, coeff)]
            last_t = np.where(acc, t, last_t)
            y = np.where(m, ny, y)
            f = np.where(m, nf, f)
            t = np.where(acc, nt, t)
            dt = np.where(active, ndt, dt)
            i = i + active
            n_att += active
            if rec:
                dt_seq.append(np.where(active & (t < ts[-1]), dt, 0.0))        # zero once the chain has reached the end
        with np.errstate(divide="ignore", invalid="ignore"):
            s = ((target - last_t) / (t - last_t))[:, None]
        a, b, c, d_, e = coeff
        outs.append((((a * s + b) * s + c) * s + d_) * s + e)
    if stats is not None:
        stats["n_attempted"] = n_att
        stats["n_evals"] = n_evals
        # the lock-step loop records one column per ROUND; a chain's attempt j is its j-th ACTIVE round: compress
        e = np.zeros((B, 0))
        stats.update(_compress_rounds(np.stack(dt_seq, 1), np.stack(acc_seq, 1) if acc_seq else e.astype(bool),
                                      np.stack(ratio_seq, 1) if ratio_seq else e, np.stack(dt_own, 1),
                                      np.stack(act_seq, 1) if act_seq else e.astype(bool), n_att))
    return np.stack(outs)


def _compress_rounds(dt_seq, acc_seq, ratio_seq, dt_own, act_seq, n_att):
    """Per chain, keep only the rounds of the lock-step loop in which it was active (a chain idles while slower ones
    still integrate, and with several output times between reaching one target and the round the slowest reaches
    it): a chain's attempt j is its j-th ACTIVE round."""
    B = dt_seq.shape[0]
    amax = int(n_att.max()) if B else 0
    out = dict(dt_seq=np.zeros((B, amax + 1)), acc_seq=np.zeros((B, amax), bool), ratio_seq=np.zeros((B, amax)),
               dt_own=np.zeros((B, amax + 1)))
    out["dt_seq"][:, 0], out["dt_own"][:, 0] = dt_seq[:, 0], dt_own[:, 0]
    for b in range(B):
        idx = np.flatnonzero(act_seq[b])
        n = len(idx)
        assert n == n_att[b]
        out["acc_seq"][b, :n] = acc_seq[b, idx]
        out["ratio_seq"][b, :n] = ratio_seq[b, idx]
        out["dt_own"][b, 1:n + 1] = dt_own[b, 1 + idx]
        out["dt_seq"][b, 1:n + 1] = dt_seq[b, 1 + idx]
    return out


def odeint_fixed(fun, y0, ts, method, nsteps, stats=None):
    """Fixed-step explicit integrator over ``[ts[0], ts[-1]]``: ``nsteps`` equal steps of classical RK4 (``method="rk4"``) or
    forward Euler (``"euler"``).  BUILD-SIDE MODE, NOT IN THE REFERENCE: the reference integrates with the adaptive Dopri5 above
    (``exe_flow_matching.py:345-349``); BASELINE.json's north star names an "RK4/Euler ODE integrator", the mode in which every
    chain takes the same number of steps.  Returns ``[len(ts), B, n]`` like ``odeint`` (intermediate output times must fall on
    step boundaries)."""
    B = y0.shape[0]
    t0, t1 = float(ts[0]), float(ts[-1])
    h = (t1 - t0) / nsteps
    marks = {int(round((float(tt) - t0) / h)): j for j, tt in enumerate(ts)}
    for j, tt in enumerate(ts):
        assert abs(t0 + h * int(round((float(tt) - t0) / h)) - float(tt)) < 1e-12, "output times must be step boundaries"
    outs = [None] * len(ts)
    y = y0.copy()
    outs[0] = y0
    for n in range(nsteps):
        t = np.full(B, t0 + n * h)
        if method == "euler":
            y = y + h * fun(y, t)
        elif method == "rk4":
            k1 = fun(y, t)
            k2 = fun(y + 0.5 * h * k1, t + 0.5 * h)
            k3 = fun(y + 0.5 * h * k2, t + 0.5 * h)
            k4 = fun(y + h * k3, t + h)
            y = y + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        else:
            raise ValueError(method)
        if n + 1 in marks:
            outs[marks[n + 1]] = y
    if stats is not None:
        stats["n_attempted"] = np.full(B, nsteps, dtype=np.int64)
        stats["n_evals"] = nsteps * (4 if method == "rk4" else 1)
    return np.stack(outs)


def _integrate(fun, y0, ts, rtol, atol, mxstep, stats, replay, fixed):
    """``fixed = (method, nsteps)`` selects the fixed-step mode; None: the reference's adaptive Dopri5."""
    if fixed:
        assert replay is None
        return odeint_fixed(fun, y0, ts, fixed[0], int(fixed[1]), stats)
    return odeint(fun, y0, ts, rtol, atol, mxstep, stats, replay)


def _augmented(model, params, z, hutch, sign, round32=False):
    """RHS of the augmented ODE.  sign=+1: ``:208-218`` (forward); sign=-1: ``:225-239`` (inverse).

    ``round32`` (a test yardstick, no counterpart in the reference): evaluate the field at the stage input ROUNDED TO FLOAT32
    -- the least any float32 implementation does to the state.  The difference it makes to a solve measures how
    well-conditioned that solve is (the clipped field of ``dim > 128`` has a log-det integrand with narrow spikes wherever
    ``|grad log pi|`` crosses the clip: there it is not)."""
    d = model.dim

    def fun(y, t):
        x = y[:, :d]
        if round32:
            x = x.astype(np.float32).astype(np.float64)
        tt = t if sign > 0 else 1.0 - t                                     # :229
        if hutch:
            v, jv = model.forward(params, x, tt, tangent=z)                 # :212-214 / :232-234
            dldj = (z * jv).sum(1)
        else:
            v = model.forward(params, x, tt)
            dldj = model.jacobian_trace(params, x, tt)                      # :216-217 / :236-237
        if sign > 0:
            return np.concatenate([v, -dldj[:, None]], axis=1)              # :218
        return np.concatenate([-v, dldj[:, None]], axis=1)                  # :230,239
    return fun


def transform_and_logdet(model, params, keys, ref_sample, hutch, rtol, atol, mxstep, n_ts=2,
                         stats=None, z=None, replay=None, round32=False, fixed=None):
    """``exe_flow_matching.py:206-221``; ``keys`` [B, 2] (one Hutchinson key per chain) or one key."""
    B, d = ref_sample.shape
    if hutch and z is None:
        keys = np.asarray(keys)
        z = prng.normal_rows(keys, d) if keys.ndim == 2 else np.broadcast_to(prng.normal(keys, (d,)), (B, d))
    y0 = np.concatenate([ref_sample, np.zeros((B, 1))], axis=1)             # :220
    ys = _integrate(_augmented(model, params, z, hutch, +1, round32), y0, np.linspace(0.0, 1.0, n_ts), rtol, atol, mxstep, stats, replay, fixed)
    return ys[-1][:, :d], ys[-1][:, d]                                      # :221


def inverse_and_logdet(model, params, keys, target_sample, hutch, rtol, atol, mxstep, n_ts=2,
                       stats=None, z=None, replay=None, round32=False, fixed=None):
    """``exe_flow_matching.py:223-242``."""
    B, d = target_sample.shape
    if hutch and z is None:
        keys = np.asarray(keys)
        z = prng.normal_rows(keys, d) if keys.ndim == 2 else np.broadcast_to(prng.normal(keys, (d,)), (B, d))
    y0 = np.concatenate([target_sample, np.zeros((B, 1))], axis=1)          # :241
    ys = _integrate(_augmented(model, params, z, hutch, -1, round32), y0, np.linspace(0.0, 1.0, n_ts), rtol, atol, mxstep, stats, replay, fixed)
    return ys[-1][:, :d], ys[-1][:, d]                                      # :242
