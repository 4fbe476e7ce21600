"""Small host-side math helpers (reference: maths/maths.py:3-33)."""
import torch


def symmetric_linear_smoothing(x, is_fuzzy, eps):
    """hard indicator 1{x>0} or its linear ramp on [-eps, eps] (maths/maths.py:3-6)"""
    if not is_fuzzy:
        return (x > 0).to(torch.float64)
    return torch.clamp((x + eps) / (2 * eps), min=0.0, max=1.0)


def compute_degree_of_truth(x, is_fuzzy, eps=0.05):
    return symmetric_linear_smoothing(x, is_fuzzy, eps)


def bisection_search(func, low: float = 1e-10, high: float = 5.0, tolerance: float = 1e-12, iters: int = 100):
    f_lo, f_hi = func(low), func(high)
    tries = 0
    while f_lo * f_hi > 0.0 and tries < 20:
        high *= 2.0
        f_hi = func(high)
        tries += 1
    if f_lo * f_hi > 0.0:
        return None
    for _ in range(iters):
        mid = 0.5 * (low + high)
        f_mid = func(mid)
        if abs(f_mid) < tolerance or (high - low) < 1e-12:
            return mid
        if f_lo * f_mid <= 0.0:
            high, f_hi = mid, f_mid
        else:
            low, f_lo = mid, f_mid
    return 0.5 * (low + high)
