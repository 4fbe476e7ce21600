#!/usr/bin/env python3
"""config 5 (Bermudan swaption EPE + PFE, 2 M paths x 120 dates, 262,144-path LSM) on one GPU: wall-clock phases and a host profile
   python tools/prof_cfg5.py [--cprofile]"""
import cProfile
import io
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import run_configs as rc
import torch
from mcx import _native

be = _native.HipBackend(0)
r = rc.config5(be)
print(json.dumps(r, default=float), flush=True)
if "--cprofile" in sys.argv:
    import numpy as np
    model = rc.VasicekModel(0.0, 0.03, 0.05, 0.1, 0.01)
    und = rc.InterestRateSwap(0.0, 16.0, 1.0, 0.03, 0.25, 0.25, rc.IRSType.PAYER)
    prod = rc.BermudanOption(und, [0.125 * k for k in range(1, 121)], 0.0, rc.OptionType.CALL)
    tl = np.array([0.125 * k for k in range(0, 121)])
    rm = rc.RiskMetrics([rc.EPEMetric(), rc.PFEMetric(0.95)], exposure_timeline=tl)
    sc = rc.SimulationController([rc.NettingSet(name="berm", products=[prod])], model, rm, 1 << 21, 1 << 18, 1, rc.SS.EULER, backend=be)
    pr = cProfile.Profile()
    pr.enable()
    sc.run_simulation()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(s.getvalue())
    print(json.dumps(sc.timings, default=float), json.dumps(getattr(sc, "prepare_timings", {}), default=float))
