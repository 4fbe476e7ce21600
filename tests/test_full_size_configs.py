"""BASELINE.json configurations 3, 4 and 5 at their FULL sizes on the GPU (run with -m gpu), through the C ABI.

Each configuration is checked three ways:
  * equal to the CPU oracle (same Philox counters) at a sub-sample the oracle finishes in seconds;
  * at full size against the statistical anchors the reference itself reports (CVA 0.004623 +- 0.000012 from
    the reference at 50 k paths; Heston semi-analytic 134.7714021047608; BS closed forms), with the z-score in the message;
  * size-independent properties (finite values, EPE >= 0, zero exposure after the last product date, full size agrees with the
    sub-sample within the Monte-Carlo error)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

REF_CVA, REF_CVA_SE = 0.004623, 0.000012        # the reference itself, config 3 at 50 k + 50 k paths
HESTON_SEMI_ANALYTIC = 134.7714021047608        # european_option.py:156-262 for the config-4 parameters


def _config3(be, n_main, n_pre):
    import bench
    return bench.build_controller(n_main, n_pre, be)


def test_config3_irs_cva_full_size_and_oracle_subsample(hip, oracle):
    # (a) 65,536 + 16,384 paths: HIP == oracle on identical counters (LSM coefficients, CVA, MC error)
    out = {}
    for be in (hip, oracle):
        sc = _config3(be, 65536, 16384)
        res = sc.run_simulation()
        out[be.name] = (np.array(res.results[0][0][0]), [c.numpy().copy() for c in sc.regression_coeffs])
    g, c = out["hip"], out["oracle"]
    assert np.isclose(g[0][0], c[0][0], rtol=1e-8), (g[0], c[0])
    assert np.isclose(g[0][1], c[0][1], rtol=1e-6), (g[0], c[0])
    for a, b in zip(g[1], c[1]):
        assert np.allclose(a, b, rtol=1e-6, atol=1e-8 * max(np.abs(b).max(), 1e-300))
    # (b) full size: 1,048,576 main + 131,072 pre-simulation paths x 250 steps
    sc = _config3(hip, 1 << 20, 131072)
    res = sc.run_simulation()
    cva, se = res.results[0][0][0]
    assert sc.timings.get("fused") is True
    z = (cva - REF_CVA) / math.hypot(se, REF_CVA_SE)
    assert math.isfinite(cva) and abs(z) < 3.0, f"CVA {cva} +- {se} vs reference {REF_CVA} +- {REF_CVA_SE}: z = {z:.2f}"
    # full size agrees with the sub-sample (independent pre-simulations: allow the LSM sampling noise, 4 joint sigma)
    assert abs(cva - g[0][0]) < 4.0 * math.hypot(se, g[0][1]) + 4e-5, (cva, g[0])
    # the one-launch plan and the materialising plan (paths tensor + evaluation pass) give the same estimator
    sc2 = _config3(hip, 1 << 20, 131072)
    sc2.main_plan = "semi"
    res2 = sc2.run_simulation()
    assert np.isclose(res2.results[0][0][0][0], cva, rtol=1e-9), (res2.results[0][0][0], cva)


def test_config4_heston_qe_full_size(hip, oracle):
    def build(be, n, diff, steps=500):
        model = cases.HestonModel(0, 800.0, 0.04, 0.45545583, -0.78975708, 0.01713417, 2.0, 0.0286834)
        prod = cases.EuropeanOption(cases.Equity(), 1.0, 720.0, cases.OptionType.CALL)
        return cases.SimulationController([cases.NettingSet(name="call", products=[prod])], model, cases.RiskMetrics([cases.PVMetric()]),
                                          n, 0, steps, cases.Q, differentiate=diff, backend=be)
    # oracle sub-sample, hard branch, identical counters
    o = {}
    for be in (hip, oracle):
        o[be.name] = np.array(build(be, 65536, False, 100).run_simulation().results[0][0][0])
    assert np.isclose(o["hip"][0], o["oracle"][0], rtol=1e-9) and np.isclose(o["hip"][1], o["oracle"][1], rtol=1e-6), o
    # full size: 4,194,304 paths x 500 QE steps, hard branch against the semi-analytic price
    pv, se = build(hip, 1 << 22, False).run_simulation().results[0][0][0]
    z = (pv - HESTON_SEMI_ANALYTIC) / se
    assert abs(z) < 4.0, f"Heston QE PV {pv} +- {se} vs semi-analytic {HESTON_SEMI_ANALYTIC}: z = {z:.2f}"
    # fuzzy branch with all seven sensitivities (differentiate=True semantics of heston.py:161-253)
    res = build(hip, 1 << 22, True).run_simulation()
    pvf, sef = res.results[0][0][0]
    greeks = res.get_derivatives(0, "pv", evaluation_idx=0)
    assert math.isfinite(pvf) and abs(pvf - HESTON_SEMI_ANALYTIC) < 1.0, (pvf, sef)      # the smoothing shifts the primal (~ +0.25)
    assert len(greeks) == 7 and all(math.isfinite(float(v)) for v in greeks.values()), greeks
    assert 0.5 < float(greeks["spot"]) < 1.0, greeks                                      # a delta


def test_config5_bermudan_swaption_full_size(hip, oracle):
    def build(be, n_main, n_pre):
        model = cases.VasicekModel(0.0, 0.03, 0.05, 0.1, 0.01)
        und = cases.InterestRateSwap(0.0, 16.0, 1.0, 0.03, 0.25, 0.25, cases.IRSType.PAYER)
        prod = cases.BermudanOption(und, [0.125 * k for k in range(1, 121)], 0.0, cases.OptionType.CALL)
        tl = np.array([0.125 * k for k in range(0, 121)])
        rm = cases.RiskMetrics([cases.EPEMetric(), cases.PFEMetric(0.95)], exposure_timeline=tl)
        return cases.SimulationController([cases.NettingSet(name="berm", products=[prod])], model, rm, n_main, n_pre, 1, cases.E, backend=be)
    o = {}
    for be in (hip, oracle):
        res = build(be, 65536, 16384).run_simulation()
        o[be.name] = (np.array(res.results[0][0]), np.array(res.results[0][1]))
    for m in (0, 1):
        assert np.allclose(o["hip"][m][:, 0], o["oracle"][m][:, 0], rtol=1e-8, atol=1e-10), (m, np.abs(o["hip"][m][:, 0] - o["oracle"][m][:, 0]).max())
        assert np.allclose(o["hip"][m][:, 1], o["oracle"][m][:, 1], rtol=1e-5, atol=1e-10), m
    # full size: 2,097,152 main + 262,144 pre-simulation paths x 120 exercise = exposure dates
    sc = build(hip, 1 << 21, 1 << 18)
    res = sc.run_simulation()
    epe, pfe = np.array(res.results[0][0]), np.array(res.results[0][1])
    assert np.isfinite(epe).all() and np.isfinite(pfe).all()
    assert (epe[:, 0] >= 0.0).all() and (epe[:, 1] >= 0.0).all()
    assert abs(epe[-1, 0]) < 1e-12 and abs(pfe[-1, 0]) < 1e-12      # nothing left after the last exercise date (controller.py:321-322)
    assert pfe[1, 0] > epe[1, 0] > 0.0                              # early dates: nearly every path still holds the option
    # full size against the sub-sample: EPE(0) (deterministic date: the regression value at t = 0) and the whole profile
    assert abs(epe[0, 0] - o["hip"][0][0, 0]) < 0.02 * abs(epe[0, 0]) + 1e-4, (epe[0], o["hip"][0][0])
    d = np.abs(epe[:, 0] - o["hip"][0][:, 0])
    tol = 5.0 * np.hypot(epe[:, 1], o["hip"][0][:, 1]) + 0.02 * np.abs(epe[:, 0]).max()
    assert (d <= tol).all(), (d.max(), tol.min())
