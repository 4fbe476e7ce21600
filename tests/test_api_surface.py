"""Host API parity with the reference's surface: constructors, error behaviour, names, results access
(reference tests: test_simulation_results_named_access.py, controller.py:40-48, 89-97)."""
import numpy as np
import pytest

import cases
from mcx.common.enums import SimulationScheme
from mcx.controller.controller import SimulationController
from mcx.controller.simulation_results import SimulationResults
from mcx.metrics.cva_metric import CVAMetric
from mcx.metrics.pfe_metric import PFEMetric
from mcx.metrics.pv_metric import PVMetric
from mcx.metrics.risk_metrics import RiskMetrics
from mcx.models.black_scholes import BlackScholesModel
from mcx.products.equity import Equity
from mcx.products.european_option import EuropeanOption
from mcx.products.netting_set import NettingSet
from mcx.products.product import OptionType


def test_constructor_errors(oracle):
    model = BlackScholesModel(0, 100.0, 0.05, 0.2)
    prod = EuropeanOption(Equity(), 1.0, 100.0, OptionType.CALL)
    ns = NettingSet(name="a", products=[prod])
    rm = RiskMetrics([PVMetric()])
    with pytest.raises(ValueError):
        SimulationController([], model, rm, 10, 0, 1, SimulationScheme.ANALYTICAL, backend=oracle)
    with pytest.raises(ValueError):
        SimulationController([ns, NettingSet(name="b", products=[prod])], model, rm, 10, 0, 1, SimulationScheme.ANALYTICAL, backend=oracle)
    with pytest.raises(Exception, match="ModelConfig"):
        SimulationController([ns], model, RiskMetrics([CVAMetric("cp", 0.4)], exposure_timeline=[0.0, 1.0]), 10, 10, 1,
                             SimulationScheme.ANALYTICAL, backend=oracle)
    with pytest.raises(ValueError):
        NettingSet(name="x", products=[])
    with pytest.raises(ValueError):
        NettingSet(name="x", products=[prod], threshold=-1.0)


def test_metric_and_param_names():
    assert PFEMetric(0.95).get_name() == "pfe[0.95]"
    assert CVAMetric("cp", 0.4).get_name() == "cva[cp]"
    ns, model, rm = cases.mixed_cva()
    assert model.get_model_param_names()[:4] == ["equity.spot", "equity.volatility", "equity.rate", "rates.rate"]
    assert SimulationController._make_unique_names(["a", "b", "a"]) == ["a", "b", "a#2"]
    assert PFEMetric(0.95).q_index(1024) == 972 and PFEMetric(0.95).q_index(1000000) == 949999


def test_simulation_results_named_and_legacy_access():
    res = SimulationResults([[[(1.0, 0.1), (2.0, 0.2)], [(3.0, 0.3)]]], [[[(0.5, 0.6), (0.7, 0.8)], [(0.9, 1.0)]]], [],
                            netting_set_names=["NS"], metric_names=["epe", "pv"], model_param_names=["spot", "vol"])
    assert np.array_equal(res.get_results("ns", "EPE"), [1.0, 2.0])
    assert res.get_results(0, 1, evaluation_idx=0) == 3.0
    assert res.get_mc_error(product="NS", metric_idx=0, evaluation_index=1) == 0.2
    assert res.get_derivatives("NS", "pv", evaluation_idx=0) == {"spot": 0.9, "vol": 1.0}
    assert res.get_derivatives("NS", "epe", param="vol", evaluation_idx=1) == 0.8
    with pytest.raises(KeyError):
        res.get_results("nope", "pv")
    with pytest.raises(TypeError):
        res.get_results("NS", "pv", bogus=1)
    with pytest.raises(ValueError):
        res.get_results("NS", "pv", prod_idx=0, product=1)


def test_analytic_pv_metric_skips_monte_carlo(oracle):
    from mcx.metrics.metric import Metric
    model = BlackScholesModel(0, 120.0, 0.05, 0.2)
    prod = EuropeanOption(Equity(), 2.0, 100.0, OptionType.CALL)
    ns = NettingSet(name="a", products=[prod])
    rm = RiskMetrics([PVMetric(evaluation_type=Metric.EvaluationType.ANALYTICAL)])
    sc = SimulationController([ns], model, rm, 1, 0, 1, SimulationScheme.ANALYTICAL, backend=oracle)
    res = sc.run_simulation()
    assert res.get_results("a", "pv", 0) == pytest.approx(31.96482, abs=1e-4)
    assert res.get_mc_error("a", "pv", 0) == 0.0


def test_analytic_pv_first_and_second_derivatives(oracle):
    """tests/pytests/test_european_option_hessian.py: PVMetric(ANALYTICAL) + differentiate + compute_higher_derivatives ->
    autograd gradient and Hessian of the Black-Scholes closed form (no simulation)"""
    import math
    from mcx.metrics.metric import Metric
    model = BlackScholesModel(0.0, 100.0, 0.05, 0.3)
    product = EuropeanOption(Equity("id"), 2.0, 100.0, OptionType.CALL)
    sc = SimulationController([NettingSet(name=product.get_name(), products=[product])], model,
                              RiskMetrics(metrics=[PVMetric(evaluation_type=Metric.EvaluationType.ANALYTICAL)]), 1, 0, 1,
                              SimulationScheme.ANALYTICAL, differentiate=True, backend=oracle)
    sc.compute_higher_derivatives()
    res = sc.run_simulation()
    assert res.get_product_names() == ["EuropeanOption"] and res.get_metric_names() == ["pv"]
    assert res.get_model_param_names() == ["spot", "volatility", "rate"]
    S, K, r, sig, T = 100.0, 100.0, 0.05, 0.3, 2.0
    d1 = (math.log(S / K) + (r + 0.5 * sig * sig) * T) / (sig * math.sqrt(T)); d2 = d1 - sig * math.sqrt(T)
    N = lambda x: 0.5 * (1 + math.erf(x / math.sqrt(2))); phi = math.exp(-0.5 * d1 * d1) / math.sqrt(2 * math.pi)
    assert res.get_results("EuropeanOption", "pv", evaluation_idx=0) == pytest.approx(S * N(d1) - K * math.exp(-r * T) * N(d2), rel=1e-12)
    g = res.get_derivatives("EuropeanOption", "pv", evaluation_idx=0)
    assert g["spot"] == pytest.approx(N(d1), rel=1e-10) and g["volatility"] == pytest.approx(S * phi * math.sqrt(T), rel=1e-10)
    assert g["rate"] == pytest.approx(K * T * math.exp(-r * T) * N(d2), rel=1e-10)
    H = res.get_second_derivatives("EuropeanOption", "pv", evaluation_idx=0)
    assert float(H["spot"]["spot"]) == pytest.approx(float(product.compute_dDeltadSpot_analytically(model)), rel=1e-9)
    assert float(H["volatility"]["volatility"]) == pytest.approx(float(product.compute_dVegadSigma_analytically(model)), rel=1e-9)
    assert float(H["spot"]["volatility"]) == pytest.approx(float(H["volatility"]["spot"]), rel=1e-12)
    assert float(H["spot"]["volatility"]) == pytest.approx(-phi * d2 / sig, rel=1e-9)          # vanna


def test_entry_points_plan_on_one_host_thread_and_restore_the_pool(oracle, monkeypatch):
    """the controller's host-side planning runs with torch's intra-op pool at one thread (mcx/helpers/host_threads.py: a pool sized
    from the core count spins a container with a CPU quota into CFS throttling) and hands the caller's setting back"""
    import torch
    before = torch.get_num_threads()
    if before == 1:
        pytest.skip("the pool is already single-threaded here")
    seen = []
    real = SimulationController._compile_all
    monkeypatch.setattr(SimulationController, "_compile_all", lambda self: (seen.append(torch.get_num_threads()), real(self))[1])
    sc, _ = cases.make_controller("bs_european", oracle, inject=False)
    assert torch.get_num_threads() == before
    sc.run_simulation()
    assert seen == [1] and torch.get_num_threads() == before
