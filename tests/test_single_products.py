"""The reference's single-product sweep (tests/pytests/test_single_product_executor_parity.py:38-240, without the gas storage):
every product family prices under `differentiate=True` through the same controller call, PV / MC error / every parameter
derivative finite; derivatives cross-checked against an independent central difference of the PV (common random numbers)."""
import copy

import numpy as np
import pytest

import cases
from mcx.aad import _set_param
from mcx.models.vasicek import VasicekModel
from mcx.products.bond import Bond
from mcx.products.swap import InterestRateSwap, IRSType

O, E_ = cases.OptionType, cases.Equity
BS = lambda: cases.BlackScholesModel(0.0, 100.0, 0.03, 0.2, asset_id="asset")
VAS = lambda: VasicekModel(0.0, 0.02, 0.03, 1.2, 0.01, asset_id="rate")
CASES = {
    "european": (BS, lambda: cases.EuropeanOption(E_("asset"), 1.0, 100.0, O.CALL, asset_id="asset"), 0),
    "binary": (BS, lambda: cases.BinaryOption(1.0, 100.0, 10.0, O.CALL, asset_id="asset"), 0),
    "basket": (lambda: cases.BlackScholesMulti(0.0, 0.03, ["asset_1", "asset_2"], [100.0, 105.0], [0.20, 0.24], np.array([[1.0, 0.35], [0.35, 1.0]])),
               lambda: cases.BasketOption(1.0, ["asset_1", "asset_2"], [0.55, 0.45], 100.0, O.CALL, cases.BasketOptionType.ARITHMETIC, False), 0),
    "barrier": (BS, lambda: cases.BarrierOption(0.0, 1.0, 100.0, 10, O.CALL, 130.0, cases.BarrierOptionType.UPANDOUT, asset_id="asset"), 256),
    "asian": (BS, lambda: cases.AsianOption(0.0, 1.0, 100.0, 10, O.CALL, cases.AsianAveragingType.ARITHMETIC, asset_id="asset"), 256),
    "american": (BS, lambda: cases.AmericanOption(E_("asset"), 1.0, 8, 100.0, O.PUT, asset_id="asset"), 256),
    "flexicall": (BS, lambda: cases.FlexiCall([cases.EuropeanOption(E_("asset"), 0.5, 95.0, O.CALL, asset_id="asset"),
                                                cases.EuropeanOption(E_("asset"), 1.0, 100.0, O.CALL, asset_id="asset"),
                                                cases.EuropeanOption(E_("asset"), 1.5, 105.0, O.CALL, asset_id="asset")], 2, asset_id="asset"), 256),
    "bond": (VAS, lambda: Bond(0.0, 1.0, 1.0, 0.5, True, 0.02, "rate"), 0),
    "swap": (VAS, lambda: InterestRateSwap(0.0, 1.0, 1.0, 0.02, 0.5, 0.5, IRSType.PAYER, "rate"), 0),
}


def _run(name, backend, model=None, diff=True):
    build_model, build_product, n_pre = CASES[name]
    product = build_product()
    product.name = name
    sc = cases.SimulationController([cases.NettingSet(name=name, products=[product])], model or build_model(),
                                    cases.RiskMetrics(metrics=[cases.PVMetric()]), 256, n_pre, 1, cases.A, differentiate=diff,
                                    backend=backend)
    return sc.run_simulation()


@pytest.mark.parametrize("name", list(CASES))
def test_single_product_pv_and_derivatives(name, oracle):
    res = _run(name, oracle)
    pv = float(res.get_results(name, "pv", evaluation_idx=0))
    assert np.isfinite(pv) and np.isfinite(float(res.get_mc_error(name, "pv", evaluation_idx=0)))
    d = res.get_derivatives(name, "pv", evaluation_idx=0)
    assert d.keys() and all(np.isfinite(float(v)) for v in d.values())
    if name in ("american", "flexicall"):
        return                       # exercise decisions flip under a bump: the derivative is only checked to be finite
    model0 = CASES[name][0]()
    for j, (pname, theta) in enumerate(zip(model0.get_model_param_names(), [float(p) for p in model0.get_model_params()])):
        h = 1e-6 * max(abs(theta), 1e-2)        # small: a wider bump moves individual paths across payoff kinks (256 paths)
        pv_pm = []
        for sgn in (1.0, -1.0):
            m = copy.deepcopy(model0)
            _set_param(m, j, theta + sgn * h)
            pv_pm.append(float(_run(name, oracle, model=m, diff=False).get_results(name, "pv", evaluation_idx=0)))
        fd = (pv_pm[0] - pv_pm[1]) / (2 * h)
        assert float(d[pname]) == pytest.approx(fd, rel=5e-3, abs=5e-4 * max(1.0, abs(pv))), (name, pname, float(d[pname]), fd)
