"""Optimizer chain: apply_if_finite(chain(adamw(mask=not-bias), clip(1.0)), 10), float32 state.

ORACLE (test infrastructure; see oracle/__init__.py).  Follows
``exe_flow_matching.py:116-137,181-186`` (optimizer), ``:189-198`` (LR schedule) and
``flax.training.train_state.TrainState.apply_gradients`` (``:366``).  The update
rules are the published ones of optax 0.1.9 (``environment.yaml:177``): PARITY
UNPINNED against optax itself; pinned against ``torch.optim.AdamW`` (same rule) in
tests/test_oracle_optim.py.  Quirks kept: ``optax.clip`` clips the
final UPDATES elementwise; the schedule is ``lr * (1 - count / learning_iter)``
evaluated at the inner optimizer's pre-increment count.
"""
import numpy as np

f32 = np.float32


def learning_rate_fn(num_train_steps, num_warmup_steps, learning_rate):
    """``exe_flow_matching.py:189-198`` (join_schedules([linear warmup, linear decay], [warmup]))."""
    def schedule(step):
        step = float(step)
        if num_warmup_steps > 0 and step < num_warmup_steps:
            frac = 1.0 - min(max(step, 0.0), num_warmup_steps) / num_warmup_steps
            return (0.0 - learning_rate) * frac + learning_rate
        ts = num_train_steps - num_warmup_steps
        if ts <= 0:
            return learning_rate
        c = min(max(step - num_warmup_steps, 0.0), ts)
        return learning_rate * (1.0 - c / ts)
    return schedule


class TrainState:
    """params + optimizer state (``exe_flow_matching.py:181-186``)."""

    def __init__(self, params, lr_fn, b1=0.9, b2=0.999, eps=1e-8, weight_decay=1e-4,
                 clip=1.0, max_consecutive_errors=10):
        self.params = [{k: v.astype(f32).copy() for k, v in p.items()} for p in params]
        self.mu = [{k: np.zeros_like(v, dtype=f32) for k, v in p.items()} for p in params]
        self.nu = [{k: np.zeros_like(v, dtype=f32) for k, v in p.items()} for p in params]
        self.lr_fn, self.b1, self.b2, self.eps, self.wd, self.clip = lr_fn, b1, b2, eps, weight_decay, clip
        self.max_err = max_consecutive_errors
        self.step = 0            # TrainState.step: incremented on every apply_gradients
        self.count = 0           # inner adam / schedule count: incremented only on accepted updates
        self.notfinite_count = 0

    def apply_gradients(self, grads):
        isfinite = all(np.isfinite(g[k]).all() for g in grads for k in g)
        self.notfinite_count = 0 if isfinite else self.notfinite_count + 1
        self.step += 1
        if not (isfinite or self.notfinite_count > self.max_err):
            return False                                  # zero update, inner state untouched
        c1 = self.count + 1
        bc1 = f32(1.0 - self.b1 ** c1)
        bc2 = f32(1.0 - self.b2 ** c1)
        lr = f32(-self.lr_fn(self.count))
        b1, b2, eps, wd = f32(self.b1), f32(self.b2), f32(self.eps), f32(self.wd)
        for p, m, v, g in zip(self.params, self.mu, self.nu, grads):
            for k in p:
                gk = g[k].astype(f32)
                m[k] = b1 * m[k] + (f32(1) - b1) * gk
                v[k] = b2 * v[k] + (f32(1) - b2) * gk * gk
                u = (m[k] / bc1) / (np.sqrt(v[k] / bc2) + eps)
                if k != "bias":                           # decay_mask_fn, :116-127
                    u = u + wd * p[k]
                u = lr * u
                u = np.clip(u, -f32(self.clip), f32(self.clip))        # optax.clip, :137
                p[k] = (p[k] + u).astype(f32)
        self.count = c1
        return True
