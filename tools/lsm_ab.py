#!/usr/bin/env python3
"""MFMA vs VALU normal-equation kernels of the LSM step (k3_step_mfma / k3_step_valu) at 262,144 pre-simulation paths, K = 3:
S = 1 (config 3: IRS exposure regression, 51 dates) and S = 2 (config 5: Bermudan swaption, 120 dates).  Run under
`rocprofv3 --kernel-trace --stats`: the per-kernel averages are the result.   python tools/lsm_ab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import run_configs as rc
from mcx import _native

be = _native.HipBackend(0)
for mfma in (False, True, False, True):
    sc = bench.build_controller(1 << 18, 1 << 18, be)
    sc.use_mfma = mfma
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sc.prepare(); torch.cuda.synchronize()
    t3 = time.perf_counter() - t0
    model = rc.VasicekModel(0.0, 0.03, 0.05, 0.1, 0.01)
    und = rc.InterestRateSwap(0.0, 16.0, 1.0, 0.03, 0.25, 0.25, rc.IRSType.PAYER)
    prod = rc.BermudanOption(und, [0.125 * k for k in range(1, 121)], 0.0, rc.OptionType.CALL)
    rm = rc.RiskMetrics([rc.EPEMetric()], exposure_timeline=np.array([0.125 * k for k in range(0, 121)]))
    sc5 = rc.SimulationController([rc.NettingSet(name="berm", products=[prod])], model, rm, 4096, 1 << 18, 1, rc.SS.EULER, backend=be)
    sc5.use_mfma = mfma
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sc5.prepare(); torch.cuda.synchronize()
    print(f"mfma={mfma}: prepare config-3 shape {t3*1e3:.2f} ms, config-5 shape {(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
