"""Host-side helpers with the reference's names (maths/maths.py): the fuzzy indicator used by payoffs / QE branching and a
bracketing root finder.  The device-side twins live in csrc/mcx_device.h (degree_of_truth)."""
import torch


def symmetric_linear_smoothing(x, is_fuzzy, eps):
    """is_fuzzy: the ramp that rises linearly from 0 at -eps to 1 at +eps; otherwise the step function 1{x > 0}"""
    if is_fuzzy:
        return ((x + eps) / (2.0 * eps)).clamp(0.0, 1.0)
    return torch.where(x > 0, torch.ones_like(x, dtype=torch.float64), torch.zeros_like(x, dtype=torch.float64))


def compute_degree_of_truth(x, is_fuzzy, eps=0.05):
    return symmetric_linear_smoothing(x, is_fuzzy, eps)


def bisection_search(func, low: float = 1e-10, high: float = 5.0, tolerance: float = 1e-12, iters: int = 100):
    """root of `func` in [low, high]; the upper end is doubled (at most 20 times) until the bracket changes sign; None if it
    never does"""
    a, b = float(low), float(high)
    fa, fb = func(a), func(b)
    for _ in range(20):
        if fa * fb <= 0.0:
            break
        b *= 2.0
        fb = func(b)
    else:
        if fa * fb > 0.0:
            return None
    for _ in range(iters):
        mid = a + 0.5 * (b - a)
        fm = func(mid)
        if abs(fm) < tolerance or (b - a) < 1e-12:
            return mid
        if (fa < 0.0) == (fm < 0.0) and fm != 0.0:
            a, fa = mid, fm
        else:
            b, fb = mid, fm
    return a + 0.5 * (b - a)
