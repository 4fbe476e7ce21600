#!/usr/bin/env python3
"""Replicates config 3 with independent Philox keys to estimate the sampling distribution of the CVA estimator
(statistical parity study against the reference's torch-RNG runs)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "montecarlo-risk-engine_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import bench
from mcx import _native

be = _native.HipBackend(0)
n_main, n_pre, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
vals = []
for t in range(reps):
    sc = bench.build_controller(n_main, n_pre, be)
    sc.seed_offset = 1000 * t
    r = sc.run_simulation()
    vals.append(r.results[0][0][0])
v = np.array([x[0] for x in vals])
print(json.dumps({"n_main": n_main, "n_pre": n_pre, "reps": reps, "mean": v.mean(), "std": v.std(ddof=1),
                  "se_of_mean": v.std(ddof=1) / np.sqrt(reps), "mean_reported_mc_error": float(np.mean([x[1] for x in vals]))}))
