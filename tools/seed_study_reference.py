#!/usr/bin/env python3
"""Reference side of the seed study (BUILD CONTAINER ONLY: imports /root/reference/src, which never travels to the GPU box).

Runs the reference's own SimulationController on BASELINE config 3 (Vasicek + CIR++ rho = 0.5 payer IRS CVA, 51 dates x 5 Euler
sub-steps) `reps` times with independent torch seeds and prints the sampling distribution of its CVA estimate.  The reference fixes
its seeds inside MonteCarloEngine.__init__ (`torch.manual_seed(42 if is_pre_simulation else 43)`, engine/engine.py:25); a
replication t runs with 42 / 43 + 1000 t by wrapping torch.manual_seed for the duration of the run (the reference's code is not
modified).  Output: one JSON line, merged by hand into profiles/r03_seed_study.json next to tools/seed_study.py's line.

  python tools/seed_study_reference.py <n_main> <n_pre> <reps> [first_t]
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "/root/reference/src")
from common.enums import SimulationScheme                        # noqa: E402
from controller.controller import SimulationController           # noqa: E402
from metrics.cva_metric import CVAMetric                         # noqa: E402
from metrics.risk_metrics import RiskMetrics                     # noqa: E402
from models.cirpp import CIRPPModel                              # noqa: E402
from models.model_config import ModelConfig                      # noqa: E402
from models.vasicek import VasicekModel                          # noqa: E402
from products.netting_set import NettingSet                      # noqa: E402
from products.swap import InterestRateSwap, IRSType              # noqa: E402

HAZARDS = {0.5: 0.006402303360855854, 1.0: 0.01553038972325307, 2.0: 0.009729741230773657, 3.0: 0.015552544648116201,
           4.0: 0.021196186202801115, 5.0: 0.02284319986706472, 7.0: 0.010111423894480876, 10.0: 0.00613267811172937,
           15.0: 0.0036969930706003337, 20.0: 0.003791311459217732}          # tests/pytests/test_cva.py:20-31


def run(n_main, n_pre, offset):
    ir = VasicekModel(0.0, rate=0.03, mean=0.05, mean_reversion_speed=0.1, volatility=0.01, asset_id="irs")
    cr = CIRPPModel(0.0, "cp", HAZARDS, kappa=0.1, theta=0.01, volatility=0.02, y0=1e-4)
    model = ModelConfig([ir, cr], inter_asset_correlation_matrix=np.array([0.5]))
    irs = InterestRateSwap(0.0, 12.5, 1.0, 0.03, 0.25, 0.25, IRSType.PAYER, "irs")
    ns = [NettingSet(name="irs", products=[irs], counterparty_id="cp")]
    rm = RiskMetrics([CVAMetric("cp", 0.4)], exposure_timeline=np.arange(51) * 0.25)
    sc = SimulationController(ns, model, rm, n_main, n_pre, 5, SimulationScheme.EULER)
    real = torch.manual_seed
    torch.manual_seed = lambda s: real(int(s) + offset)
    try:
        res = sc.run_simulation()
    finally:
        torch.manual_seed = real
    v, e = res.results[0][0][0]
    return float(v), float(e)


if __name__ == "__main__":
    n_main, n_pre, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    t0 = time.time()
    vals = [run(n_main, n_pre, 1000 * t) for t in range(first, first + reps)]
    v = np.array([x[0] for x in vals])
    print(json.dumps({"n_main": n_main, "n_pre": n_pre, "reps": reps, "seeds": f"42/43 + 1000 t, t = {first}..{first + reps - 1}",
                      "mean": v.mean(), "std": v.std(ddof=1) if reps > 1 else None,
                      "se_of_mean": v.std(ddof=1) / np.sqrt(reps) if reps > 1 else None,
                      "mean_reported_mc_error": float(np.mean([x[1] for x in vals])), "seconds": time.time() - t0,
                      "torch": torch.__version__, "threads": torch.get_num_threads()}))
