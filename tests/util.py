"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# BASELINE.json north_star: "within 1e-3 relative fp32 tolerance".  The metric used
# everywhere: max|a-b| <= RTOL * max|b|  (relative to the tensor's scale).
RTOL = 1e-3


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_err(actual, expected):
    a = actual.detach().cpu().double().numpy() if torch.is_tensor(actual) else np.asarray(actual, dtype=np.float64)
    e = expected.detach().cpu().double().numpy() if torch.is_tensor(expected) else np.asarray(expected, dtype=np.float64)
    assert a.shape == e.shape, (a.shape, e.shape)
    scale = np.abs(e).max()
    if scale == 0:
        return float(np.abs(a).max())
    return float(np.abs(a - e).max() / scale)


def check(actual, fx, name, tol=RTOL):
    """Compare against a packed golden entry (channel-strided if it was large)."""
    step = int(fx[name + "__cstep"]) if (name + "__cstep") in fx.files else 1
    a = actual.detach().cpu()
    if step > 1:
        a = a[:, ::step]
    err = rel_err(a, fx[name])
    assert err <= tol, "%s: rel err %.3e > %.1e" % (name, err, tol)
    return err


def check_grad(p_grad, fx, key, tol=RTOL):
    if key + "__first8" in fx.files:
        err = rel_err(p_grad[:8], fx[key + "__first8"])
    else:
        exp = fx[key]
        g = p_grad
        if exp.shape != tuple(g.shape):
            g = g[:8]
        err = rel_err(g, exp)
    assert err <= tol, "%s: rel err %.3e > %.1e" % (key, err, tol)
    return err
