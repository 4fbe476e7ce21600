"""The controller cases of tests/golden/gen_golden.py rebuilt with the mcx classes (same constructor arguments)."""
import os

import numpy as np
import torch

from mcx.common.enums import SimulationScheme
from mcx.controller.controller import SimulationController
from mcx.metrics.ce_metric import CEMetric
from mcx.metrics.cva_metric import CVAMetric
from mcx.metrics.eepe_metric import EEPEMetric
from mcx.metrics.ene_metric import ENEMetric
from mcx.metrics.epe_metric import EPEMetric
from mcx.metrics.pfe_metric import PFEMetric
from mcx.metrics.pv_metric import PVMetric
from mcx.metrics.risk_metrics import RiskMetrics
from mcx.models.black_scholes import BlackScholesModel
from mcx.models.black_scholes_multi import BlackScholesMulti
from mcx.models.cirpp import CIRPPModel
from mcx.models.heston import HestonModel
from mcx.models.model_config import ModelConfig
from mcx.models.vasicek import VasicekModel
from mcx.products.basket_option import BasketOption, BasketOptionType
from mcx.products.bermudan_option import AmericanOption, BermudanOption
from mcx.products.binary_option import BinaryOption
from mcx.products.flexicall import FlexiCall
from mcx.products.barrier_option import BarrierOption, BarrierOptionType
from mcx.products.asian_option import AsianOption, AsianAveragingType
from mcx.products.bond import Bond
from mcx.products.equity import Equity
from mcx.products.european_option import EuropeanOption
from mcx.products.netting_set import NettingSet
from mcx.products.product import OptionType
from mcx.products.swap import InterestRateSwap, IRSType

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
A, E, Q = SimulationScheme.ANALYTICAL, SimulationScheme.EULER, SimulationScheme.QE

HAZARDS = {
    0.5: 0.006402303360855854, 1.0: 0.01553038972325307, 2.0: 0.009729741230773657,
    3.0: 0.015552544648116201, 4.0: 0.021196186202801115, 5.0: 0.02284319986706472,
    7.0: 0.010111423894480876, 10.0: 0.00613267811172937, 15.0: 0.0036969930706003337,
    20.0: 0.003791311459217732,
}


def bs_european():
    model = BlackScholesModel(0, 120.0, 0.05, 0.2)
    prod = EuropeanOption(Equity(), 2.0, 100.0, OptionType.CALL)
    return [NettingSet(name=prod.get_name(), products=[prod])], model, RiskMetrics([PVMetric()])


def s2f_european():
    from mcx.models.schwartz_two_factor import SchwartzTwoFactorModel
    model = SchwartzTwoFactorModel(0.0, [0.0, 0.5, 1.0, 2.0], [30.0, 32.0, 31.0, 29.0], rate=0.03, short_term_mean_reversion=1.5,
                                   short_term_vol=0.4, long_term_drift=0.01, long_term_vol=0.15, rho=0.3, asset_id="gas")
    prod = EuropeanOption(Equity("gas"), 1.0, 30.0, OptionType.CALL)
    return [NettingSet(name=prod.get_name(), products=[prod])], model, RiskMetrics([PVMetric()])


def bs_put_euler():
    model = BlackScholesModel(0, 90.0, 0.03, 0.35)
    prod = EuropeanOption(Equity(), 1.5, 100.0, OptionType.PUT)
    return [NettingSet(name=prod.get_name(), products=[prod])], model, RiskMetrics([PVMetric()])


def irs_models(rho, a=0.1, sig=0.01):
    ir = VasicekModel(0.0, 0.03, 0.05, a, sig, asset_id="irs")
    cr = CIRPPModel(0.0, "cp", HAZARDS, kappa=0.1, theta=0.01, volatility=0.02, y0=1e-4)
    return ModelConfig([ir, cr], inter_asset_correlation_matrix=np.array([rho]))


def irs_cva():
    model = irs_models(0.5)
    irs = InterestRateSwap(0.0, 2.5, 1.0, 0.03, 0.25, 0.25, IRSType.PAYER, "irs")
    ns = [NettingSet(name=irs.get_name(), products=[irs], counterparty_id="cp")]
    return ns, model, RiskMetrics([CVAMetric("cp", 0.4)], exposure_timeline=np.arange(11) * 0.25)


def irs_cva_linspace():
    model = irs_models(0.99999, a=0.02, sig=0.2)
    irs = InterestRateSwap(0.0, 2.0, 1.0, 0.03, 0.25, 0.25, IRSType.PAYER, "irs")
    ns = [NettingSet(name=irs.get_name(), products=[irs], counterparty_id="cp")]
    return ns, model, RiskMetrics([CVAMetric("cp", 0.4), PVMetric(), EPEMetric(), ENEMetric()],
                                  exposure_timeline=np.linspace(0, 2.0, 7))


def zcb_cva():
    ir = VasicekModel(0.0, 0.03, 0.05, 1.0, 0.2, asset_id="bond")
    cr = CIRPPModel(0.0, "cp", HAZARDS, kappa=0.1, theta=0.01, volatility=0.02, y0=1e-4)
    model = ModelConfig([ir, cr], inter_asset_correlation_matrix=np.array([0.0]))
    zb = Bond(0.0, 2.0, 1, 2.0, True, 0.0, "bond")
    ns = [NettingSet(name=zb.get_name(), products=[zb], counterparty_id="cp")]
    return ns, model, RiskMetrics([CVAMetric("cp", 0.4)], exposure_timeline=np.linspace(0, 2.0, 9))


def heston():
    model = HestonModel(0, 800.0, 0.04, 0.45545583, -0.78975708, 0.01713417, 2.0, 0.0286834)
    prod = EuropeanOption(Equity(), 1.0, 720.0, OptionType.CALL)
    return [NettingSet(name=prod.get_name(), products=[prod])], model, RiskMetrics([PVMetric()])


def heston_highpsi():
    model = HestonModel(0, 100.0, 0.02, 1.2, -0.5, 0.5, 0.02, 0.02)
    prod = EuropeanOption(Equity(), 1.0, 95.0, OptionType.PUT)
    return [NettingSet(name=prod.get_name(), products=[prod])], model, RiskMetrics([PVMetric()])


def bermudan_swaption():
    model = VasicekModel(0.0, 0.03, 0.05, 0.1, 0.01)
    und = InterestRateSwap(0.0, 2.0, 1.0, 0.03, 0.25, 0.25, IRSType.PAYER)
    prod = BermudanOption(und, [0.125 * k for k in range(1, 16)], 0.0, OptionType.CALL)
    ns = [NettingSet(name="berm_ns", products=[prod])]
    tl = np.array([0.125 * k for k in range(0, 17)])
    mets = [EPEMetric(), PFEMetric(0.95), ENEMetric(), EEPEMetric(), CEMetric(), PVMetric(), PFEMetric(0.5)]
    return ns, model, RiskMetrics(mets, exposure_timeline=tl)


def american():
    model = BlackScholesModel(0.0, 100.0, 0.05, 0.5)
    prod = AmericanOption(Equity("id"), 3.0, 12, 100.0, OptionType.PUT)
    return [NettingSet(name=prod.get_name(), products=[prod])], model, RiskMetrics([PVMetric()])


def netting():
    model = VasicekModel(0.0, 0.02, 0.04, 0.3, 0.015, asset_id="r")
    irs1 = InterestRateSwap(0.0, 2.0, 1.0, 0.025, 0.5, 0.25, IRSType.PAYER, "r")
    irs2 = InterestRateSwap(0.0, 1.5, 2.0, 0.035, 0.5, 0.5, IRSType.RECEIVER, "r")
    bond = Bond(0.0, 2.0, 0.1, 0.5, True, 0.03, "r")
    frn = Bond(0.0, 1.0, 0.05, 0.25, True, None, "r")
    ns1 = NettingSet(name="ns_coll", products=[irs1, bond], threshold=0.002, margin_period_of_risk=0.25)
    ns2 = NettingSet(name="ns_thr", products=[irs2, frn], threshold=0.01)
    tl = np.array([0.0, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0])
    mets = [EPEMetric(), ENEMetric(), PFEMetric(0.9), PVMetric(), EEPEMetric(), CEMetric()]
    return [ns1, ns2], model, RiskMetrics(mets, exposure_timeline=tl)


def bond_option():
    model = VasicekModel(0.0, 0.03, 0.05, 0.2, 0.02, asset_id="r")
    und_b = Bond(0.0, 3.0, 1.0, 0.5, True, 0.04, "r")
    und_s = InterestRateSwap(0.0, 3.0, 1.0, 0.035, 0.5, 0.25, IRSType.PAYER, "r")
    o1 = EuropeanOption(und_b, 1.0, 0.98, OptionType.CALL, asset_id="r")
    o1.name = "bond_call"
    o2 = EuropeanOption(und_s, 1.0, 0.0, OptionType.PUT, asset_id="r")
    o2.name = "swaption_put"
    return [NettingSet(name="o1", products=[o1]), NettingSet(name="o2", products=[o2])], model, RiskMetrics([PVMetric()])


def mixed_cva():
    eq = BlackScholesModel(0.0, 100.0, 0.03, 0.22, asset_id="equity")
    ra = VasicekModel(0.0, 0.03, 0.03, 1.0, 0.01, asset_id="rates")
    cr = CIRPPModel(0.0, "cp", HAZARDS, kappa=0.10, theta=0.01, volatility=0.02, y0=1e-4, deterministic=True)
    model = ModelConfig([eq, ra, cr], inter_asset_correlation_matrix=[np.array([0.0])] * 3)
    prods = []
    o = EuropeanOption(Equity("equity"), 1.0, 95.0, OptionType.CALL, asset_id="equity"); o.name = "call0"; prods.append(o)
    o = EuropeanOption(Equity("equity"), 2.0, 105.0, OptionType.CALL, asset_id="equity"); o.name = "call1"; prods.append(o)
    b = Bond(0.0, 2.0, 2.0, 0.5, True, 0.02, "rates"); b.name = "bond0"; prods.append(b)
    s = InterestRateSwap(0.0, 2.0, 25.0, 0.025, 0.5, 0.25, IRSType.PAYER, "rates"); s.name = "swap0"; prods.append(s)
    ns = [NettingSet(name="mixed", products=prods, counterparty_id="cp")]
    return ns, model, RiskMetrics([CVAMetric("cp", 0.4), EPEMetric()], exposure_timeline=np.linspace(0.0, 2.5, 8))


def _baskets(ids, cv):
    b = BasketOption(1.0, ids, [0.4, 0.3, 0.2, 0.1], 100, OptionType.CALL, BasketOptionType.ARITHMETIC, cv); b.name = "basket_arithmetic"
    g = BasketOption(1.0, ids, [0.4, 0.3, 0.2, 0.1], 95, OptionType.PUT, BasketOptionType.GEOMETRIC); g.name = "basket_geometric"
    return [NettingSet(name=b.get_name(), products=[b]), NettingSet(name=g.get_name(), products=[g])]


def basket_model_config():
    ids = ["asset1", "asset2", "asset3", "asset4"]
    models = [BlackScholesModel(0.0, 100.0 + 5 * k, 0.02, 0.4 - 0.05 * k, asset_id=ids[k]) for k in range(4)]
    model = ModelConfig(models=models, inter_asset_correlation_matrix=np.array([[0.5], [0.3], [0.1], [0.5], [0.2], [0.4]]))
    return _baskets(ids, False), model, RiskMetrics([PVMetric()])


def basket_multi():
    ids = ["asset1", "asset2", "asset3", "asset4"]
    corr = np.array([[1.0, 0.5, 0.3, 0.1], [0.5, 1.0, 0.5, 0.2], [0.3, 0.5, 1.0, 0.4], [0.1, 0.2, 0.4, 1.0]])
    model = BlackScholesMulti(0.0, 0.02, ids, [100.0, 105.0, 110.0, 115.0], [0.4, 0.35, 0.3, 0.25], corr)
    return _baskets(ids, True), model, RiskMetrics([PVMetric()])


def binary_asian():
    model = BlackScholesModel(0, 100.0, 0.03, 0.25)
    prods = [BinaryOption(1.0, 100.5, 10.0, OptionType.CALL), BinaryOption(0.75, 99.0, 5.0, OptionType.PUT),
             AsianOption(0.0, 1.0, 100.0, 5, OptionType.CALL, AsianAveragingType.ARITHMETIC),
             AsianOption(0.25, 1.25, 102.0, 5, OptionType.PUT, AsianAveragingType.GEOMETRIC)]
    for k, p in enumerate(prods):
        p.name = f"p{k}"
    return [NettingSet(name=p.name, products=[p]) for p in prods], model, RiskMetrics([PVMetric()])


def barrier():
    model = BlackScholesModel(0, 100.0, 0.03, 0.25)
    B = BarrierOptionType
    prods = [BarrierOption(0.0, 1.0, 100.0, 6, OptionType.CALL, 125.0, B.UPANDOUT),
             BarrierOption(0.0, 1.0, 105.0, 6, OptionType.PUT, 90.0, B.DOWNANDIN),
             BarrierOption(0.2, 1.2, 95.0, 5, OptionType.CALL, 85.0, B.DOWNANDOUT, 130.0, B.UPANDOUT),
             BarrierOption(0.0, 0.8, 100.0, 5, OptionType.PUT, 110.0, B.UPANDIN, 80.0, B.DOWNANDOUT)]
    for k, p in enumerate(prods):
        p.name = f"b{k}"
    return [NettingSet(name=p.name, products=[p]) for p in prods], model, RiskMetrics([PVMetric()])


def flexicall():
    model = BlackScholesModel(0, 100.0, 0.03, 0.25)
    opts = [EuropeanOption(Equity(), 0.25 * (k + 1), 98.0 + 2.0 * k, OptionType.PUT) for k in range(5)]
    fc = FlexiCall(opts, 2); fc.name = "flexi"
    opts3 = [EuropeanOption(Equity(), 0.2 * (k + 1), 101.0 - k, OptionType.CALL) for k in range(4)]
    fc3 = FlexiCall(opts3, 3); fc3.name = "flexi3"
    tl = np.array([0.0, 0.25, 0.5, 0.75, 1.0, 1.25])
    return [NettingSet(name="flexi", products=[fc]), NettingSet(name="flexi3", products=[fc3])], model, \
        RiskMetrics([PVMetric(), EPEMetric()], exposure_timeline=tl)


def _mixed_book_products(mod):
    """every product family of the reference's large-book script (pv_performance_large_netting_set.py:86-233), a few of each;
    `mod` supplies the classes (the reference's modules here, mcx's in tests/cases.py)"""
    ids = [f"asset_{k}" for k in range(4)]
    P = []
    O, E = mod["OptionType"], mod["Equity"]
    for i in range(6):
        a = ids[i % 4]
        P.append(mod["EuropeanOption"](E(a), [0.25, 0.5, 0.75, 1.0, 1.5, 2.0][i], [80.0, 90.0, 100.0, 110.0, 120.0][i % 5], O.CALL if i % 2 == 0 else O.PUT, asset_id=a))
    for i in range(3):
        p = mod["BinaryOption"]([0.5, 1.0, 1.5][i], [90.0, 100.0, 110.0][i], 8.0 + 2.0 * i, O.CALL if i % 2 == 0 else O.PUT, asset_id=ids[i % 4]); p.name = f"binary_{i}"; P.append(p)
    bw = [[0.5, 0.3, 0.2, 0.0], [0.25, 0.25, 0.25, 0.25], [0.4, 0.35, 0.15, 0.10]]
    for i in range(3):
        k = 2 + i
        w = bw[i][:k]
        p = mod["BasketOption"]([0.75, 1.25, 2.0][i], ids[:k], [x / sum(w) for x in w], 95.0 + 5.0 * i, O.CALL if i % 2 == 0 else O.PUT,
                                mod["BasketOptionType"].ARITHMETIC if i != 0 else mod["BasketOptionType"].GEOMETRIC, False); p.name = f"basket_{i}"; P.append(p)
    for i in range(3):
        p = mod["AsianOption"](0.0, [0.5, 1.0, 1.5][i], 88.0 + 6.0 * i, [8, 12, 18][i], O.CALL if i % 2 == 0 else O.PUT,
                               mod["AsianAveragingType"].ARITHMETIC if i != 0 else mod["AsianAveragingType"].GEOMETRIC, asset_id=ids[(i + 1) % 4]); p.name = f"asian_{i}"; P.append(p)
    for i in range(4):
        p = mod["BarrierOption"](0.0, [0.5, 0.75, 1.25, 1.75][i], 85.0 + 7.5 * i, [8, 12, 18, 24][i], O.CALL if i % 3 != 0 else O.PUT,
                                 [118.0, 125.0, 132.0, 140.0][i] + 2.0 * (i % 2), mod["BarrierOptionType"].UPANDOUT, asset_id=ids[i % 4]); p.name = f"barrier_{i}"; P.append(p)
    for i in range(3):
        a = ids[(i + 2) % 4]
        p = mod["AmericanOption"](E(a), [0.75, 1.0, 1.5][i], [8, 12, 18][i], [80.0, 100.0, 120.0][i], O.PUT if i % 2 == 0 else O.CALL, asset_id=a); p.name = f"american_{i}"; P.append(p)
    for i in range(3):
        a = ids[i % 4]
        mat, L = [1.0, 1.5, 2.0][i], [3, 4, 5][i]
        und = [mod["EuropeanOption"](E(a), float(t), 90.0 + 6.0 * ((i + k) % 6), O.CALL, asset_id=a) for k, t in enumerate(np.linspace(mat / L, mat, L))]
        p = mod["FlexiCall"](und, min(1 + i, L - 1), asset_id=a); p.name = f"flexicall_{i}"; P.append(p)
    return ids, P


def mixed_book_multi():
    ids, P = _mixed_book_products(globals())
    corr = np.full((4, 4), 0.35); np.fill_diagonal(corr, 1.0)
    market = BlackScholesMulti(0.0, 0.03, ids, [95.0 + 7.5 * k for k in range(4)], [0.18 + 0.03 * k for k in range(4)], corr)
    credit = CIRPPModel(0.0, "cp", HAZARDS, kappa=0.10, theta=0.01, volatility=0.02, y0=0.0001, deterministic=False)
    model = ModelConfig([market, credit], inter_asset_correlation_matrix=[np.full((4, 1), 0.2)])
    horizon = max(float(p.modeling_timeline[-1]) for p in P)
    ns = [NettingSet(name="book", products=P, counterparty_id="cp", margin_period_of_risk=10 / 252)]
    return ns, model, RiskMetrics([CVAMetric("cp", 0.4), EPEMetric()], exposure_timeline=np.linspace(0.0, horizon, 12))


def barrier_bridge():
    ns, model, rm = barrier()
    for n in ns:
        n.products[0].set_use_brownian_bridge()
    return ns, model, rm


def bs_european_exposure():
    model = BlackScholesModel(0, 100.0, 0.03, 0.25)
    c = EuropeanOption(Equity(), 1.0, 95.0, OptionType.CALL); c.name = "call"
    p = EuropeanOption(Equity(), 1.5, 110.0, OptionType.PUT); p.name = "put"
    ns = [NettingSet(name="opts", products=[c, p], threshold=1.0),
          NettingSet(name="put_only", products=[EuropeanOption(Equity(), 0.75, 100.0, OptionType.PUT)])]
    tl = np.array([0.0, 0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 2.0])
    return ns, model, RiskMetrics([EPEMetric(), PFEMetric(0.9), PVMetric()], exposure_timeline=tl)


# name -> (builder, n_pre, n_main, num_steps, scheme, differentiate)
CASES = {
    "bs_european": (bs_european, 0, 2048, 10, A, False),
    "bs_european_aad": (bs_european, 0, 2048, 10, A, True),
    "bs_put_euler_aad": (bs_put_euler, 0, 1024, 8, E, True),
    "irs_cva": (irs_cva, 1024, 1024, 2, E, False),
    "irs_cva_linspace": (irs_cva_linspace, 1024, 1024, 2, E, False),
    "zcb_cva": (zcb_cva, 1024, 1024, 2, E, False),
    "heston_qe": (heston, 0, 1024, 8, Q, False),
    "heston_qe_aad": (heston, 0, 1024, 8, Q, True),
    "heston_highpsi_qe": (heston_highpsi, 0, 1024, 6, Q, False),
    "heston_euler": (heston, 0, 1024, 8, E, False),
    "bermudan_swaption": (bermudan_swaption, 1024, 1024, 1, E, False),
    "american_put": (american, 2048, 1024, 1, A, False),
    "netting": (netting, 1024, 1024, 1, A, False),
    "bond_option": (bond_option, 0, 1024, 2, A, False),
    "mixed_cva": (mixed_cva, 512, 512, 2, E, False),
    "bs_european_exposure": (bs_european_exposure, 0, 1024, 2, A, False),
    "basket_model_config": (basket_model_config, 0, 1024, 2, A, False),
    "basket_model_config_euler": (basket_model_config, 0, 1024, 4, E, False),
    "basket_multi": (basket_multi, 0, 1024, 2, A, False),
    "basket_multi_euler": (basket_multi, 0, 1024, 3, E, False),
    "binary_asian": (binary_asian, 0, 1024, 2, A, False),
    "binary_asian_euler": (binary_asian, 0, 1024, 3, E, False),
    "barrier": (barrier, 0, 2048, 2, A, False),
    "barrier_euler": (barrier, 0, 2048, 3, E, False),
    "barrier_bridge": (barrier_bridge, 0, 2048, 2, A, False),
    "flexicall": (flexicall, 2048, 1024, 1, A, False),
    "mixed_book_multi": (mixed_book_multi, 128, 128, 1, E, False),
    "s2f_european": (s2f_european, 0, 1024, 4, A, False),
    "s2f_european_euler": (s2f_european, 0, 1024, 6, E, False),
    # sensitivities through the LSM regression; the fixtures hold only the reference gradients, draws = the base case's
    "irs_cva_aad": (irs_cva, 1024, 1024, 2, E, True),
    "mixed_cva_aad": (mixed_cva, 512, 512, 2, E, True),
    # exercise products: the reference's tape holds the exercise policy fixed (bermudan_option.py:122-128)
    "bermudan_swaption_aad": (bermudan_swaption, 1024, 1024, 1, E, True),
    "american_put_aad": (american, 2048, 1024, 1, A, True),
}
DRAWS_FROM = {"irs_cva_aad": "irs_cva", "mixed_cva_aad": "mixed_cva", "bermudan_swaption_aad": "bermudan_swaption",
              "american_put_aad": "american_put"}


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def make_controller(name, backend, inject=True, fused=True):
    build, n_pre, n_main, steps, scheme, diff = CASES[name]
    ns, model, rm = build()
    sc = SimulationController(ns, model, rm, n_main, n_pre, steps, scheme, differentiate=diff, backend=backend)
    sc.materialize = True
    sc.allow_fused = fused
    g = load_golden(name)
    gd = load_golden(DRAWS_FROM.get(name, name))
    if inject:
        def prep(key_z, key_u):
            z = backend.from_numpy(np.ascontiguousarray(np.transpose(gd[key_z], (0, 2, 1))))      # [S][n_z][N]
            u = backend.from_numpy(np.ascontiguousarray(gd[key_u][:, :, 0])) if key_u in gd.files else None
            return z, u
        sc._inject["main"] = prep("z_main", "u_main")
        bridge = {}
        for k in [k for k in g.files if k.startswith("bridge_u_")]:       # recorded numpy uniforms [calls][N][intervals]
            calls = g[k]
            rows = np.zeros((2 * calls.shape[2], calls.shape[1]))
            for bi in range(calls.shape[0]):
                rows[bi::2] = calls[bi].T
            bridge[int(k.rsplit("_", 1)[1])] = backend.from_numpy(rows)
        if bridge:
            sc._inject["bridge_main"] = bridge
        if "z_pre" in gd.files:
            sc._inject["pre"] = prep("z_pre", "u_pre")
    return sc, g
