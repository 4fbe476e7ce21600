"""times of the select entry points on a [121][2^21] matrix (BASELINE config 5's exposure shape): the bracket pass with an empty /
typical / everything-inside bracket, one digit pass; GB/s against the ~6 TB/s tools/ubench_hbm reaches for a read-only pass"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "montecarlo-risk-engine_amd"))
from mcx import _native
from mcx.plan import UnsecuredSpec

be = _native.HipBackend(0)
E, n = 121, 1 << 21
x = torch.randn(E, n, dtype=torch.float64, device="cuda")
unsec = UnsecuredSpec(np.arange(E), None, 0.0, False)
cap = 1 << 17
cand = torch.empty(E, cap, dtype=torch.float64, device="cuda")
counts = torch.zeros(2, E, dtype=torch.int64, device="cuda")
gb = E * n * 8 / 1e9


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = {}
for name, (l, h) in {"empty_bracket": (9.0, 9.5), "typical_1.7pct": (1.60, 1.70), "everything_inside": (-99.0, 99.0)}.items():
    lo = torch.full((E,), l, dtype=torch.float64, device="cuda"); hi = torch.full((E,), h, dtype=torch.float64, device="cuda")
    ms = timed(lambda: be.select_bracket(unsec, x, lo, hi, counts, cand))
    out[name] = dict(ms=ms, tb_s=gb / ms, inside=int((counts[1][0] & ((1 << 44) - 1)).item()))
prefix = torch.zeros(E, 3, dtype=torch.int64, device="cuda")
hist = torch.zeros(E, 3, 2048, dtype=torch.int64, device="cuda")
ms = timed(lambda: be.select_hist_dev(unsec, x, 3, prefix, 53, 11, hist))
out["digit_pass"] = dict(ms=ms, tb_s=gb / ms)
print(json.dumps(out, indent=1))
