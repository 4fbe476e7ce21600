#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the Python reference (read-only, /root/reference).

Run in the build container only:   python tests/golden/gen_golden.py
The reference never travels to the GPU box; only the .npz files written next to this script do.
Every case records (a) the exact torch CPU draws the reference consumed (torch.randn / torch.rand_like, in call order),
(b) the tensors the reference produced (paths, coefficients, cashflows, exposures, metric values, AAD gradients).
Our oracle / HIP path replay the draws ("inject-Z") and must reproduce (b).

Reference entry points exercised (file:line):
  engine/engine.py:27-123, models/*.py step functions, controller/controller.py:272-383 (LSM), :385-471 (evaluation),
  metrics/*.py, products/netting_set.py:48-184.
"""
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("MCX_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "src"))

import numpy as np
import torch

from common.enums import SimulationScheme
from controller.controller import SimulationController
from engine.engine import MonteCarloEngine
from metrics.ce_metric import CEMetric
from metrics.cva_metric import CVAMetric
from metrics.eepe_metric import EEPEMetric
from metrics.ene_metric import ENEMetric
from metrics.epe_metric import EPEMetric
from metrics.pfe_metric import PFEMetric
from metrics.pv_metric import PVMetric
from metrics.risk_metrics import RiskMetrics
from models.black_scholes import BlackScholesModel
from models.black_scholes_multi import BlackScholesMulti
from models.cirpp import CIRPPModel
from models.heston import HestonModel
from models.model_config import ModelConfig
from models.vasicek import VasicekModel
from products.basket_option import BasketOption, BasketOptionType
from products.bermudan_option import AmericanOption, BermudanOption
from products.binary_option import BinaryOption
from products.flexicall import FlexiCall
from products.barrier_option import BarrierOption, BarrierOptionType
from products.asian_option import AsianOption, AsianAveragingType
from products.product import OptionType as _OT  # noqa: F401
from products.bond import Bond
from products.equity import Equity
from products.european_option import EuropeanOption, OptionType
from products.netting_set import NettingSet
from products.swap import InterestRateSwap, IRSType
from request_interface.request_types import AtomicRequest, AtomicRequestType

OUT = os.path.dirname(os.path.abspath(__file__))
F64 = torch.float64

HAZARDS = {
    0.5: 0.006402303360855854, 1.0: 0.01553038972325307, 2.0: 0.009729741230773657,
    3.0: 0.015552544648116201, 4.0: 0.021196186202801115, 5.0: 0.02284319986706472,
    7.0: 0.010111423894480876, 10.0: 0.00613267811172937, 15.0: 0.0036969930706003337,
    20.0: 0.003791311459217732,
}


# ----------------------------------------------------------------------------------------------
# draw recorder: wraps torch.randn / torch.rand_like while a reference run is in progress
# ----------------------------------------------------------------------------------------------
class DrawRecorder:
    def __init__(self):
        self.normals = []
        self.uniforms = []
        self._randn = torch.randn
        self._rand_like = torch.rand_like

    def __enter__(self):
        def randn(*a, **k):
            z = self._randn(*a, **k)
            self.normals.append(z.detach().clone().numpy())
            return z

        def rand_like(*a, **k):
            u = self._rand_like(*a, **k)
            self.uniforms.append(u.detach().clone().numpy())
            return u

        torch.randn = randn
        torch.rand_like = rand_like
        return self

    def __exit__(self, *exc):
        torch.randn = self._randn
        torch.rand_like = self._rand_like

    def split(self, n_first_calls_paths, n_second_calls_paths):
        """Split recorded normals into pre-sim and main-sim blocks by path count."""
        pre = [z for z in self.normals if z.shape[0] == n_first_calls_paths]
        return pre


def npf(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def run_controller_case(name, build, n_pre, n_main, num_steps, scheme, differentiate=False, extra=None):
    """Run one SimulationController case under the recorder and dump everything."""
    netting_sets, model, risk_metrics = build()
    captured = {"paths": [], "eval": []}

    orig_gen = MonteCarloEngine.generate_paths
    orig_eval = SimulationController._evaluate_product

    def gen(self):
        p = orig_gen(self)
        captured["paths"].append(npf(p))
        return p

    def ev(self, product, resolved):
        r = orig_eval(self, product, resolved)
        captured["eval"].append({k: npf(v) for k, v in r.items()})
        return r

    MonteCarloEngine.generate_paths = gen
    SimulationController._evaluate_product = ev
    try:
        sc = SimulationController(
            netting_sets=netting_sets, model=model, risk_metrics=risk_metrics,
            num_paths_mainsim=n_main, num_paths_presim=n_pre, num_steps=num_steps,
            simulation_scheme=scheme, differentiate=differentiate,
        )
        with DrawRecorder() as rec:
            res = sc.run_simulation()
    finally:
        MonteCarloEngine.generate_paths = orig_gen
        SimulationController._evaluate_product = orig_eval

    out = {}
    out["simulation_timeline"] = npf(sc.simulation_timeline)
    out["exposure_timeline"] = npf(sc.exposure_timeline)
    out["metric_exposure_timeline"] = npf(sc.metric_exposure_timeline)
    n_paths_runs = len(captured["paths"])
    if n_paths_runs == 2:
        out["paths_pre"] = captured["paths"][0]
        out["paths_main"] = captured["paths"][1]
    else:
        out["paths_main"] = captured["paths"][0]
    # draws: split in call order. generate_paths is called pre (if any) then main.
    normals = rec.normals
    uniforms = rec.uniforms
    n_sub = None
    if n_paths_runs == 2:
        # each engine consumes the same number of randn calls
        half = len(normals) // 2
        out["z_pre"] = np.stack(normals[:half], axis=0)      # [S, N_pre, n_z]
        out["z_main"] = np.stack(normals[half:], axis=0)     # [S, N_main, n_z]
        if uniforms:
            hu = len(uniforms) // 2
            out["u_pre"] = np.stack(uniforms[:hu], axis=0)
            out["u_main"] = np.stack(uniforms[hu:], axis=0)
    else:
        out["z_main"] = np.stack(normals, axis=0)
        if uniforms:
            out["u_main"] = np.stack(uniforms, axis=0)
    for i, c in enumerate(sc.regression_coeffs):
        out[f"expo_coeffs_{i}"] = npf(c)
    for i, p in enumerate(sc.products):
        if p.regression_coeffs is not None and p.regression_coeffs.numel() > 0:
            out[f"prod_coeffs_{i}"] = npf(p.regression_coeffs)
    for i, e in enumerate(captured["eval"]):
        out[f"cfs_{i}"] = e["discounted_cashflows"]
        out[f"exposures_{i}"] = e["exposure_profiles"]
    # metric results [ns][metric][eval] -> (value, err)
    for ns_i, ns in enumerate(res.results):
        for m_i, m in enumerate(ns):
            vals = np.array([[float(v[0]), float(v[1])] for v in m])
            out[f"result_{ns_i}_{m_i}"] = vals
    if differentiate:
        for ns_i, ns in enumerate(res.derivatives):
            for m_i, m in enumerate(ns):
                g = np.array([[np.nan if d is None else float(d) for d in ev_] for ev_ in m])
                out[f"grad_{ns_i}_{m_i}"] = g
        out["param_names"] = np.array(res.model_param_names)
    out["metric_names"] = np.array(res.metric_names)
    out["netting_set_names"] = np.array(res.netting_set_names)
    if extra:
        out.update(extra(sc, model))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {os.path.getsize(path)/1024:.0f} KiB; results:",
          {k: v[:2].tolist() for k, v in out.items() if k.startswith("result_")})


# ----------------------------------------------------------------------------------------------
# A. single-step maps and request closed forms
# ----------------------------------------------------------------------------------------------
def gen_steps():
    g = torch.Generator().manual_seed(7)
    N = 256
    out = {}

    def rn(*s):
        return torch.randn(*s, dtype=F64, generator=g)

    def ru(*s):
        return torch.rand(*s, dtype=F64, generator=g)

    t1 = torch.tensor([0.4], dtype=F64)
    t2 = torch.tensor([0.65], dtype=F64)
    dt = t2 - t1
    out["t1"], out["t2"] = npf(t1), npf(t2)

    # Black-Scholes
    bs = BlackScholesModel(0.0, 120.0, 0.05, 0.2)
    S = 120.0 * torch.exp(0.3 * rn(N, 1))
    z = rn(N, 1)
    L = bs.get_cholesky(SimulationScheme.ANALYTICAL, dt)
    out["bs_chol_analytical"] = npf(L)
    out["bs_state"], out["bs_z"] = npf(S), npf(z)
    out["bs_exact"] = npf(bs.simulate_time_step_analytically(t1, t2, S, z @ L.T))
    out["bs_euler"] = npf(bs.simulate_time_step_euler(t1, t2, S, z))

    # Vasicek
    va = VasicekModel(0.0, 0.03, 0.05, 0.1, 0.01, asset_id="irs")
    st = torch.stack([0.03 + 0.02 * rn(N), 0.05 * ru(N)], dim=-1)
    z = rn(N, 1)
    L = va.get_cholesky(SimulationScheme.ANALYTICAL, dt)
    out["va_chol_analytical"] = npf(L)
    out["va_state"], out["va_z"] = npf(st), npf(z)
    out["va_exact"] = npf(va.simulate_time_step_analytically(t1, t2, st, z @ L.T))
    out["va_euler"] = npf(va.simulate_time_step_euler(t1, t2, st, z))
    # closed forms
    out["va_zcb_0_2"] = npf(va.compute_bond_price(va.calibration_date, torch.tensor(2.0, dtype=F64), st[:, 0]))
    out["va_zcb_065_3"] = npf(va.compute_bond_price(0.65, 3.0, st[:, 0]))
    req = AtomicRequest(AtomicRequestType.LIBOR_RATE, 0.4, 0.65)
    out["va_libor_04_065"] = npf(va.resolve_request(req, "irs", st))
    req = AtomicRequest(AtomicRequestType.NUMERAIRE, 0.65)
    out["va_numeraire"] = npf(va.resolve_request(req, "irs", st))

    # CIR++
    ci = CIRPPModel(0.0, "cp", HAZARDS, kappa=0.1, theta=0.01, volatility=0.02, y0=1e-4)
    st = torch.stack([torch.abs(0.01 + 0.01 * rn(N)) * (ru(N) > 0.1) - 1e-4 * (ru(N) > 0.9), 0.05 * ru(N)], dim=-1)
    z = rn(N, 1)
    out["ci_state"], out["ci_z"] = npf(st), npf(z)
    out["ci_euler"] = npf(ci.simulate_time_step_euler(t1, t2, st, z))
    times = np.array([0.0, 0.05, 0.25, 0.4, 0.5, 0.75, 1.0, 1.5, 2.0, 2.5, 3.7, 5.0, 6.2, 9.99, 10.0, 12.5, 19.0, 20.0, 25.0])
    out["ci_times"] = times
    out["ci_psi"] = np.array([float(ci.psi(torch.tensor(t, dtype=F64))) for t in times])
    out["ci_pd"] = np.array([float(ci.cs_helper.probability_of_default(ci.hazard_rates, ci.tenors, torch.tensor(t, dtype=F64)))
                             for t in times])
    pairs = [(0.0, 0.25), (0.25, 0.5), (0.4, 0.65), (1.0, 1.25), (2.0, 10.0), (12.25, 12.5), (0.0, 2.0)]
    out["ci_pairs"] = np.array(pairs)
    out["ci_cond_survival"] = np.stack([npf(ci.survival_probability(a, b, st[:, 0])) for a, b in pairs], axis=0)
    req = AtomicRequest(AtomicRequestType.SURVIVAL_PROBABILITY)
    out["ci_survival"] = npf(ci.resolve_request(req, "cp", st))
    # deterministic CIR++ (cirpp.py:155-172)
    cd = CIRPPModel(0.0, "cp", HAZARDS, kappa=0.1, theta=0.01, volatility=0.02, y0=1e-4, deterministic=True)
    out["cd_euler"] = npf(cd.simulate_time_step_euler(t1, t2, st, z))
    out["cd_cond_survival"] = np.stack([npf(cd.survival_probability(a, b, st[:, 0])) for a, b in pairs], axis=0)
    out["cd_init_state"] = npf(cd.get_state(2))

    # Heston
    he = HestonModel(0.0, 800.0, 0.04, 0.45545583, -0.78975708, 0.01713417, 2.0, 0.0286834)
    st = torch.stack([np.log(800.0) + 0.2 * rn(N), torch.abs(0.03 + 0.05 * rn(N)) * (ru(N) > 0.15)], dim=-1)
    z = rn(N, 2)
    L = he.get_cholesky(SimulationScheme.EULER, None)
    out["he_chol_euler"] = npf(L)
    out["he_state"], out["he_z"] = npf(st), npf(z)
    out["he_euler"] = npf(he.simulate_time_step_euler(t1, t2, st, z @ L.T))
    for smooth, tag in ((False, "hard"), (True, "fuzzy")):
        he.perform_smoothing = smooth
        with DrawRecorder() as rec:
            torch.manual_seed(11)
            res = he.simulate_time_step_qe(t1, t2, st, z)
        out[f"he_qe_{tag}"] = npf(res)
        out[f"he_qe_{tag}_u"] = rec.uniforms[0]
    # a high-psi regime (small mean variance, high vol-of-vol): exercises the exponential branch
    he2 = HestonModel(0.0, 100.0, 0.02, 1.2, -0.5, 0.5, 0.02, 0.02)
    st2 = torch.stack([np.log(100.0) + 0.2 * rn(N), torch.abs(0.01 * rn(N)) * (ru(N) > 0.3)], dim=-1)
    out["he2_state"] = npf(st2)
    for smooth, tag in ((False, "hard"), (True, "fuzzy")):
        he2.perform_smoothing = smooth
        with DrawRecorder() as rec:
            torch.manual_seed(12)
            res = he2.simulate_time_step_qe(t1, t2, st2, z)
        out[f"he2_qe_{tag}"] = npf(res)
        out[f"he2_qe_{tag}_u"] = rec.uniforms[0]

    # ModelConfig Cholesky factors (model_config.py:101-142, model.py:50-73)
    for rho in (-0.95, 0.0, 0.5, 0.99999):
        mc = ModelConfig([VasicekModel(0.0, 0.03, 0.05, 0.1, 0.01, asset_id="irs"),
                          CIRPPModel(0.0, "cp", HAZARDS, 0.1, 0.01, 0.02, 1e-4)],
                         inter_asset_correlation_matrix=np.array([rho]))
        out[f"mc_chol_{rho}"] = npf(mc.get_cholesky(SimulationScheme.EULER, None))
    models = [BlackScholesModel(0.0, 100.0, 0.0, 0.4, asset_id=f"a{i}") for i in range(4)]
    mc = ModelConfig(models, inter_asset_correlation_matrix=np.array([[0.5]] * 6))
    out["mc4_chol_euler"] = npf(mc.get_cholesky(SimulationScheme.EULER, None))
    out["mc4_chol_analytical"] = npf(mc.get_cholesky(SimulationScheme.ANALYTICAL, torch.tensor([0.25], dtype=F64)))

    path = os.path.join(OUT, "steps.npz")
    np.savez_compressed(path, **out)
    print("steps:", os.path.getsize(path) // 1024, "KiB")


# ----------------------------------------------------------------------------------------------
# B. engine-only path generation (ModelConfig of 4 correlated BS, exact + Euler)
# ----------------------------------------------------------------------------------------------
def gen_paths_mc4():
    out = {}
    for scheme, tag, steps in ((SimulationScheme.ANALYTICAL, "exact", 2), (SimulationScheme.EULER, "euler", 5)):
        models = [BlackScholesModel(0.0, 100.0 + 5 * i, 0.01 * i, 0.2 + 0.1 * i, asset_id=f"a{i}") for i in range(4)]
        mc = ModelConfig(models, inter_asset_correlation_matrix=np.array([[0.5], [0.3], [-0.2], [0.1], [0.4], [0.6]]))
        tl = torch.tensor([0.0, 0.5, 1.0, 1.75], dtype=F64)
        eng = MonteCarloEngine(tl, scheme, mc, 512, steps, False)
        with DrawRecorder() as rec:
            p = eng.generate_paths()
        out[f"{tag}_paths"] = npf(p)
        out[f"{tag}_z"] = np.stack(rec.normals, axis=0)
        out[f"{tag}_timeline"] = npf(tl)
    path = os.path.join(OUT, "paths_mc4.npz")
    np.savez_compressed(path, **out)
    print("paths_mc4:", os.path.getsize(path) // 1024, "KiB")


# ----------------------------------------------------------------------------------------------
# C. controller cases (shrunk BASELINE configs)
# ----------------------------------------------------------------------------------------------
def case_bs_european():
    model = BlackScholesModel(0, 120.0, 0.05, 0.2)
    prod = EuropeanOption(Equity(), 2.0, 100.0, OptionType.CALL)
    pv = PVMetric()
    return [NettingSet(name=prod.get_name(), products=[prod])], model, RiskMetrics([pv])


def case_bs_put_euler():
    model = BlackScholesModel(0, 90.0, 0.03, 0.35)
    prod = EuropeanOption(Equity(), 1.5, 100.0, OptionType.PUT)
    pv = PVMetric()
    return [NettingSet(name=prod.get_name(), products=[prod])], model, RiskMetrics([pv])


def _irs_models(rho, a=0.1, sig=0.01):
    ir = VasicekModel(0.0, 0.03, 0.05, a, sig, asset_id="irs")
    cr = CIRPPModel(0.0, "cp", HAZARDS, kappa=0.1, theta=0.01, volatility=0.02, y0=1e-4)
    return ModelConfig([ir, cr], inter_asset_correlation_matrix=np.array([rho]))


def case_irs_cva():
    model = _irs_models(0.5)
    irs = InterestRateSwap(0.0, 2.5, 1.0, 0.03, 0.25, 0.25, IRSType.PAYER, "irs")
    ns = [NettingSet(name=irs.get_name(), products=[irs], counterparty_id="cp")]
    tl = np.arange(11) * 0.25
    return ns, model, RiskMetrics([CVAMetric("cp", 0.4)], exposure_timeline=tl)


def case_irs_cva_linspace():
    """reference test parameters (tests/pytests/test_cva.py:122-172) on a shrunk horizon; linspace dates interleave payments"""
    model = _irs_models(0.99999, a=0.02, sig=0.2)
    irs = InterestRateSwap(0.0, 2.0, 1.0, 0.03, 0.25, 0.25, IRSType.PAYER, "irs")
    ns = [NettingSet(name=irs.get_name(), products=[irs], counterparty_id="cp")]
    tl = np.linspace(0, 2.0, 7)
    return ns, model, RiskMetrics([CVAMetric("cp", 0.4), PVMetric(), EPEMetric(), ENEMetric()], exposure_timeline=tl)


def case_zcb_cva():
    ir = VasicekModel(0.0, 0.03, 0.05, 1.0, 0.2, asset_id="bond")
    cr = CIRPPModel(0.0, "cp", HAZARDS, kappa=0.1, theta=0.01, volatility=0.02, y0=1e-4)
    model = ModelConfig([ir, cr], inter_asset_correlation_matrix=np.array([0.0]))
    zb = Bond(0.0, 2.0, 1, 2.0, True, 0.0, "bond")
    ns = [NettingSet(name=zb.get_name(), products=[zb], counterparty_id="cp")]
    tl = np.linspace(0, 2.0, 9)
    return ns, model, RiskMetrics([CVAMetric("cp", 0.4)], exposure_timeline=tl)


def case_heston():
    model = HestonModel(0, 800.0, 0.04, 0.45545583, -0.78975708, 0.01713417, 2.0, 0.0286834)
    prod = EuropeanOption(Equity(), 1.0, 720.0, OptionType.CALL)
    return [NettingSet(name=prod.get_name(), products=[prod])], model, RiskMetrics([PVMetric()])


def case_heston_highpsi():
    model = HestonModel(0, 100.0, 0.02, 1.2, -0.5, 0.5, 0.02, 0.02)
    prod = EuropeanOption(Equity(), 1.0, 95.0, OptionType.PUT)
    return [NettingSet(name=prod.get_name(), products=[prod])], model, RiskMetrics([PVMetric()])


def case_bermudan_swaption():
    model = VasicekModel(0.0, 0.03, 0.05, 0.1, 0.01)
    und = InterestRateSwap(0.0, 2.0, 1.0, 0.03, 0.25, 0.25, IRSType.PAYER)
    ex = [0.125 * k for k in range(1, 16)]
    prod = BermudanOption(und, ex, 0.0, OptionType.CALL)
    ns = [NettingSet(name="berm_ns", products=[prod])]
    tl = np.array([0.125 * k for k in range(0, 17)])
    mets = [EPEMetric(), PFEMetric(0.95), ENEMetric(), EEPEMetric(), CEMetric(), PVMetric(), PFEMetric(0.5)]
    return ns, model, RiskMetrics(mets, exposure_timeline=tl)


def case_american():
    model = BlackScholesModel(0.0, 100.0, 0.05, 0.5)
    prod = AmericanOption(Equity("id"), 3.0, 12, 100.0, OptionType.PUT)
    return [NettingSet(name=prod.get_name(), products=[prod])], model, RiskMetrics([PVMetric()])


def case_netting():
    """two netting sets on one Vasicek model: threshold + MPoR collateral (netting_set.py:48-184)"""
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


def case_bond_option():
    """European option on a coupon bond and on a swap under Vasicek (european_option.py:45-68, bond.py:115-163)"""
    model = VasicekModel(0.0, 0.03, 0.05, 0.2, 0.02, asset_id="r")
    und_b = Bond(0.0, 3.0, 1.0, 0.5, True, 0.04, "r")
    und_s = InterestRateSwap(0.0, 3.0, 1.0, 0.035, 0.5, 0.25, IRSType.PAYER, "r")
    o1 = EuropeanOption(und_b, 1.0, 0.98, OptionType.CALL, asset_id="r")
    o1.name = "bond_call"
    o2 = EuropeanOption(und_s, 1.0, 0.0, OptionType.PUT, asset_id="r")
    o2.name = "swaption_put"
    ns = [NettingSet(name="o1", products=[o1]), NettingSet(name="o2", products=[o2])]
    return ns, model, RiskMetrics([PVMetric()])


def case_mixed_cva():
    """BS + Vasicek + deterministic CIR++ book (tests/exposure_tests/cva_large_netting_set_derivatives.py:133-171), tiny"""
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
    tl = np.linspace(0.0, 2.5, 8)
    return ns, model, RiskMetrics([CVAMetric("cp", 0.4), EPEMetric()], exposure_timeline=tl)


def _baskets(ids, cv):
    b = BasketOption(1.0, ids, [0.4, 0.3, 0.2, 0.1], 100, OptionType.CALL, BasketOptionType.ARITHMETIC, cv); b.name = "basket_arithmetic"
    g = BasketOption(1.0, ids, [0.4, 0.3, 0.2, 0.1], 95, OptionType.PUT, BasketOptionType.GEOMETRIC); g.name = "basket_geometric"
    return [NettingSet(name=b.get_name(), products=[b]), NettingSet(name=g.get_name(), products=[g])]


def case_basket_model_config():
    """4 correlated Black-Scholes models in a ModelConfig + arithmetic / geometric baskets (tests/pytests/test_model_config.py:18-71)"""
    ids = ["asset1", "asset2", "asset3", "asset4"]
    models = [BlackScholesModel(0.0, 100.0 + 5 * k, 0.02, 0.4 - 0.05 * k, asset_id=ids[k]) for k in range(4)]
    model = ModelConfig(models=models, inter_asset_correlation_matrix=np.array([[0.5], [0.3], [0.1], [0.5], [0.2], [0.4]]))
    return _baskets(ids, False), model, RiskMetrics([PVMetric()])


def case_basket_multi():
    """BlackScholesMulti + arithmetic basket with the geometric control variate (tests/pytests/test_pv_basket_option.py:16-69)"""
    ids = ["asset1", "asset2", "asset3", "asset4"]
    corr = np.array([[1.0, 0.5, 0.3, 0.1], [0.5, 1.0, 0.5, 0.2], [0.3, 0.5, 1.0, 0.4], [0.1, 0.2, 0.4, 1.0]])
    model = BlackScholesMulti(0.0, 0.02, ids, [100.0, 105.0, 110.0, 115.0], [0.4, 0.35, 0.3, 0.25], corr)
    return _baskets(ids, True), model, RiskMetrics([PVMetric()])


def case_binary_asian():
    """fuzzy binary payoffs (binary_option.py:38-43) and Asian averages incl. the first-date numeraire (asian_option.py:76-90)"""
    model = BlackScholesModel(0, 100.0, 0.03, 0.25)
    prods = [BinaryOption(1.0, 100.5, 10.0, OptionType.CALL), BinaryOption(0.75, 99.0, 5.0, OptionType.PUT),
             AsianOption(0.0, 1.0, 100.0, 5, OptionType.CALL, AsianAveragingType.ARITHMETIC),
             AsianOption(0.25, 1.25, 102.0, 5, OptionType.PUT, AsianAveragingType.GEOMETRIC)]
    for k, p in enumerate(prods):
        p.name = f"p{k}"
    return [NettingSet(name=p.name, products=[p]) for p in prods], model, RiskMetrics([PVMetric()])


def case_barrier():
    """discretely monitored single / double barriers with fuzzy indicators (barrier_option.py:60-125)"""
    model = BlackScholesModel(0, 100.0, 0.03, 0.25)
    B = BarrierOptionType
    prods = [BarrierOption(0.0, 1.0, 100.0, 6, OptionType.CALL, 125.0, B.UPANDOUT),
             BarrierOption(0.0, 1.0, 105.0, 6, OptionType.PUT, 90.0, B.DOWNANDIN),
             BarrierOption(0.2, 1.2, 95.0, 5, OptionType.CALL, 85.0, B.DOWNANDOUT, 130.0, B.UPANDOUT),
             BarrierOption(0.0, 0.8, 100.0, 5, OptionType.PUT, 110.0, B.UPANDIN, 80.0, B.DOWNANDOUT)]
    for k, p in enumerate(prods):
        p.name = f"b{k}"
    return [NettingSet(name=p.name, products=[p]) for p in prods], model, RiskMetrics([PVMetric()])


def case_flexicall():
    """k-right exercise machines (flexicall.py:104-143): LSM over (k+1) states, exposures via the regression"""
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


def case_mixed_book_multi():
    """BlackScholesMulti(4) + stochastic CIR++ credit, every payoff family, 10-day MPoR collateral, CVA
    (tests/exposure_tests/cva_perfprmance_large_netting_set.py:69-148, shrunk)"""
    ids, P = _mixed_book_products(globals())
    corr = np.full((4, 4), 0.35); np.fill_diagonal(corr, 1.0)
    market = BlackScholesMulti(0.0, 0.03, ids, [95.0 + 7.5 * k for k in range(4)], [0.18 + 0.03 * k for k in range(4)], corr)
    credit = CIRPPModel(0.0, "cp", HAZARDS, kappa=0.10, theta=0.01, volatility=0.02, y0=0.0001, deterministic=False)
    model = ModelConfig([market, credit], inter_asset_correlation_matrix=[np.full((4, 1), 0.2)])
    horizon = max(float(p.modeling_timeline[-1]) for p in P)
    ns = [NettingSet(name="book", products=P, counterparty_id="cp", margin_period_of_risk=10 / 252)]
    return ns, model, RiskMetrics([CVAMetric("cp", 0.4), EPEMetric()], exposure_timeline=np.linspace(0.0, horizon, 12))


class _RecordingRng:
    """wraps a product's numpy Generator and keeps every uniform block it hands out (Brownian-bridge draws)"""

    def __init__(self, rng):
        self.rng, self.calls = rng, []

    def uniform(self, *a, **k):
        u = self.rng.uniform(*a, **k)
        self.calls.append(np.array(u, dtype=np.float64))
        return u


def case_barrier_bridge():
    """barrier options with the Brownian-bridge crossing correction (barrier_option.py:126-223); the uniforms of the products'
    numpy Generators are recorded and injected on the other side"""
    model = BlackScholesModel(0, 100.0, 0.03, 0.25)
    B = BarrierOptionType
    prods = [BarrierOption(0.0, 1.0, 100.0, 6, OptionType.CALL, 125.0, B.UPANDOUT),
             BarrierOption(0.0, 1.0, 105.0, 6, OptionType.PUT, 90.0, B.DOWNANDIN),
             BarrierOption(0.2, 1.2, 95.0, 5, OptionType.CALL, 85.0, B.DOWNANDOUT, 130.0, B.UPANDOUT),
             BarrierOption(0.0, 0.8, 100.0, 5, OptionType.PUT, 110.0, B.UPANDIN, 80.0, B.DOWNANDOUT)]
    for k, p in enumerate(prods):
        p.name = f"b{k}"
        p.set_use_brownian_bridge()
        p.rng = _RecordingRng(p.rng)
    return [NettingSet(name=p.name, products=[p]) for p in prods], model, RiskMetrics([PVMetric()])


def _bridge_extra(sc, model):
    return {f"bridge_u_{i}": np.stack(p.rng.calls, axis=0) for i, p in enumerate(sc.products) if isinstance(getattr(p, "rng", None), _RecordingRng)}


def case_bs_european_exposure():
    """analytic Black-Scholes exposure path (european_option.py:123-145, controller.py:430-437): no regression"""
    model = BlackScholesModel(0, 100.0, 0.03, 0.25)
    c = EuropeanOption(Equity(), 1.0, 95.0, OptionType.CALL); c.name = "call"
    p = EuropeanOption(Equity(), 1.5, 110.0, OptionType.PUT); p.name = "put"
    ns = [NettingSet(name="opts", products=[c, p], threshold=1.0), NettingSet(name="put_only", products=[EuropeanOption(Equity(), 0.75, 100.0, OptionType.PUT)])]
    tl = np.array([0.0, 0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 2.0])
    return ns, model, RiskMetrics([EPEMetric(), PFEMetric(0.9), PVMetric()], exposure_timeline=tl)


def case_s2f_european():
    """Schwartz two-factor commodity model (models/schwartz_two_factor.py:147-196): European call on the spot"""
    from models.schwartz_two_factor import SchwartzTwoFactorModel
    model = SchwartzTwoFactorModel(0.0, [0.0, 0.5, 1.0, 2.0], [30.0, 32.0, 31.0, 29.0], rate=0.03, short_term_mean_reversion=1.5,
                                   short_term_vol=0.4, long_term_drift=0.01, long_term_vol=0.15, rho=0.3, asset_id="gas")
    prod = EuropeanOption(Equity("gas"), 1.0, 30.0, OptionType.CALL)
    return [NettingSet(name=prod.get_name(), products=[prod])], model, RiskMetrics([PVMetric()])


def gen_netting_interp():
    """NettingSet collateral profile WITHOUT exact delayed indices (netting_set.py:76-107, 151-157): 'linear' and 'previous'
    interpolation of the netted exposure at t - MPoR, with and without a threshold"""
    g = torch.Generator().manual_seed(11)
    tl = torch.tensor([0.0, 0.25, 0.5, 1.0, 1.5, 2.5], dtype=torch.float64)
    expo = torch.randn(6, 64, generator=g, dtype=torch.float64) * 3.0
    prod = EuropeanOption(Equity("eq"), 1.0, 100.0, OptionType.CALL)
    out = {"timeline": tl.numpy(), "expo": expo.numpy()}
    for mode in ("linear", "previous"):
        for thr in (0.0, 0.75):
            for mpor in (0.1, 0.25, 0.6):
                ns = NettingSet(name="c", products=[prod], threshold=thr, margin_period_of_risk=mpor, collateral_interpolation=mode)
                key = f"{mode}_{thr}_{mpor}"
                out["coll_" + key] = npf(ns.compute_collateral_profile(expo, tl))
                out["unsec_" + key] = npf(ns.compute_unsecured_exposure_profiles(expo, tl))
    path = os.path.join(OUT, "netting_interp.npz")
    np.savez_compressed(path, **out)
    print("netting_interp: wrote", path)


def slim_to_gradients(name):
    """a differentiate=True fixture of a case whose draws / paths are already held by the base fixture: keep the results,
    the reference's autograd gradients and the names only"""
    path = os.path.join(OUT, name + ".npz")
    g = np.load(path)
    keep = {k: g[k] for k in g.files if k.startswith(("result_", "grad_")) or k in ("param_names", "metric_names", "netting_set_names")}
    np.savez_compressed(path, **keep)
    print(f"{name}: slimmed to {os.path.getsize(path)/1024:.0f} KiB")


NEW_CASES = {
    # round 2: Schwartz two-factor; sensitivities of exercise products (the tape freezes the exercise decisions, bermudan_option.py:122-128)
    "s2f_european": lambda: run_controller_case("s2f_european", case_s2f_european, 0, 1024, 4, SimulationScheme.ANALYTICAL),
    "s2f_european_euler": lambda: run_controller_case("s2f_european_euler", case_s2f_european, 0, 1024, 6, SimulationScheme.EULER),
    "bermudan_swaption_aad": lambda: (run_controller_case("bermudan_swaption_aad", case_bermudan_swaption, 1024, 1024, 1, SimulationScheme.EULER, differentiate=True),
                                      slim_to_gradients("bermudan_swaption_aad")),
    "american_put_aad": lambda: (run_controller_case("american_put_aad", case_american, 2048, 1024, 1, SimulationScheme.ANALYTICAL, differentiate=True),
                                 slim_to_gradients("american_put_aad")),
    "netting_interp": gen_netting_interp,
}


def main():
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "only":           # python gen_golden.py only <case> [<case> ...]
        for n in sys.argv[2:]:
            NEW_CASES[n]()
        return
    gen_steps()
    gen_paths_mc4()
    A, E, Q = SimulationScheme.ANALYTICAL, SimulationScheme.EULER, SimulationScheme.QE
    run_controller_case("bs_european", case_bs_european, 0, 2048, 10, A)
    run_controller_case("bs_european_aad", case_bs_european, 0, 2048, 10, A, differentiate=True)
    run_controller_case("bs_put_euler_aad", case_bs_put_euler, 0, 1024, 8, E, differentiate=True)
    run_controller_case("irs_cva", case_irs_cva, 1024, 1024, 2, E)
    run_controller_case("irs_cva_linspace", case_irs_cva_linspace, 1024, 1024, 2, E)
    run_controller_case("zcb_cva", case_zcb_cva, 1024, 1024, 2, E)
    run_controller_case("heston_qe", case_heston, 0, 1024, 8, Q)
    run_controller_case("heston_qe_aad", case_heston, 0, 1024, 8, Q, differentiate=True)
    run_controller_case("heston_highpsi_qe", case_heston_highpsi, 0, 1024, 6, Q)
    run_controller_case("heston_euler", case_heston, 0, 1024, 8, E)
    run_controller_case("bermudan_swaption", case_bermudan_swaption, 1024, 1024, 1, E)
    run_controller_case("american_put", case_american, 2048, 1024, 1, A)
    run_controller_case("netting", case_netting, 1024, 1024, 1, A)
    run_controller_case("bond_option", case_bond_option, 0, 1024, 2, A)
    run_controller_case("mixed_cva", case_mixed_cva, 512, 512, 2, E)
    run_controller_case("bs_european_exposure", case_bs_european_exposure, 0, 1024, 2, A)
    # sensitivities through the LSM regression (controller.py:609-627)
    run_controller_case("irs_cva_aad", case_irs_cva, 1024, 1024, 2, E, differentiate=True)
    run_controller_case("mixed_cva_aad", case_mixed_cva, 512, 512, 2, E, differentiate=True)
    # baskets: ModelConfig of n Black-Scholes models, BlackScholesMulti
    run_controller_case("basket_model_config", case_basket_model_config, 0, 1024, 2, A)
    run_controller_case("basket_model_config_euler", case_basket_model_config, 0, 1024, 4, E)
    run_controller_case("basket_multi", case_basket_multi, 0, 1024, 2, A)
    run_controller_case("basket_multi_euler", case_basket_multi, 0, 1024, 3, E)
    run_controller_case("binary_asian", case_binary_asian, 0, 1024, 2, A)
    run_controller_case("binary_asian_euler", case_binary_asian, 0, 1024, 3, E)
    run_controller_case("barrier", case_barrier, 0, 2048, 2, A)
    run_controller_case("barrier_euler", case_barrier, 0, 2048, 3, E)
    run_controller_case("flexicall", case_flexicall, 2048, 1024, 1, A)
    run_controller_case("mixed_book_multi", case_mixed_book_multi, 128, 128, 1, E)
    run_controller_case("barrier_bridge", case_barrier_bridge, 0, 2048, 2, A, extra=_bridge_extra)
    for fn in NEW_CASES.values():
        fn()


if __name__ == "__main__":
    main()
