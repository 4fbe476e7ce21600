"""The multi-path book kernels (k2_eval_book_v<PPL, FEAT>, k4_cva_paths_v) only run at path counts that fill the chip, far above
the golden fixtures' 1-2 k paths.  Here the golden BOOKS are run unfused at 0.5-1 M paths (ragged: not a multiple of any tile)
on the GPU and on the CPU oracle with identical Philox counters; cashflows, exposures and every metric must agree.  One book per
event family the kernel is specialised for: plain cashflows / exposures (four paths per lane), per-term numeraires, exercise
products, basket / binary payoffs, analytic Black-Scholes exposures.  (profiles/r02_book_variants_kernel_stats.csv lists the
kernels this module launched.)"""
import os
import sys

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from mcx.controller.controller import SimulationController

pytestmark = pytest.mark.gpu

LIGHT = (1 << 20) + 333          # >= 4096 blocks: k2_eval_book_v<4, .> and k4_cva_paths_v<2>
WIDE = (1 << 19) + 333           # >= 2048 blocks: k2_eval_book_v<2, .>
# books with a hard indicator in front of the compared tensors: the exercise decision `immediate > continuation` of Bermudan /
# American / FlexiCall products (bermudan_option.py:93-131) may flip on a last-bit difference of the regression coefficients for
# a vanishing fraction of paths; every other book's cashflows and exposures must agree entry by entry
FLIP_BOOKS = {"bermudan_swaption", "american_put", "flexicall"}
BOOKS = [("irs_cva", LIGHT), ("bond_option", LIGHT), ("netting", WIDE), ("mixed_cva", WIDE), ("bermudan_swaption", WIDE),
         ("american_put", WIDE), ("flexicall", WIDE), ("basket_multi", WIDE), ("binary_asian", WIDE), ("bs_european_exposure", WIDE)]


def _run(name, n_main, backend):
    build, n_pre, _, steps, scheme, diff = cases.CASES[name]
    ns, model, rm = build()
    sc = SimulationController(ns, model, rm, n_main, n_pre, steps, scheme, differentiate=diff, backend=backend)
    sc.materialize = True
    sc.allow_fused = False
    res = sc.run_simulation()
    return sc, res


@pytest.mark.parametrize("name,n_main", BOOKS, ids=[b[0] for b in BOOKS])
def test_multi_path_book_kernels_match_oracle(name, n_main, hip, oracle):
    sg, rg = _run(name, n_main, hip)
    so, ro = _run(name, n_main, oracle)
    for key in ("cfs", "expo"):
        a, b = sg.last_state[key], so.last_state[key]
        if a is None:
            assert b is None
            continue
        a, b = a.cpu().numpy(), b.numpy()
        bad = ~np.isclose(a, b, rtol=1e-9, atol=1e-11)
        assert bad.mean() <= (1e-4 if name in FLIP_BOOKS else 0.0), (name, key, bad.mean(), np.abs(a - b).max())
    for ns_i in range(len(sg.netting_sets)):
        for m_i, metric in enumerate(sg.risk_metrics.metrics):
            x = np.array(rg.results[ns_i][m_i], dtype=np.float64)
            y = np.array(ro.results[ns_i][m_i], dtype=np.float64)
            assert np.allclose(x[:, 0], y[:, 0], rtol=1e-7, atol=1e-10), (name, metric.get_name(), x[:, 0], y[:, 0])


def test_unfused_plan_matches_the_one_launch_plan_at_full_size(hip):
    """config 3 at 2^20 paths: K1 + k2_eval_book_v<4> + k4_cva_paths_v<2> against the one-launch kernel (same counters)"""
    import bench
    out = []
    for fused in (True, False):
        sc = bench.build_controller(1 << 20, 16384, hip)
        sc.allow_fused = fused
        out.append(sc.run_simulation().results[0][0][0])
    assert np.isclose(out[0][0], out[1][0], rtol=1e-9) and np.isclose(out[0][1], out[1][1], rtol=1e-7), out
