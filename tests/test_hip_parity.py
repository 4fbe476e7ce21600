"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI of libmcx_hip.so.

 1. inject-Z: the draws recorded from the reference are replayed on the GPU; paths, LSM coefficients, cashflows,
    exposures and every metric must match the reference's golden tensors (same tolerances as the oracle's own pin).
 2. Philox: same seeds/counters on GPU and CPU oracle -> paths agree to <= 1e-11 relative (libm ulp differences only),
    metrics to <= 1e-9; plus RNG known-answer vectors.
 3. kernels against each other: MFMA vs VALU normal equations, radix select vs sort."""
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
NON_AAD = [n for n, c in cases.CASES.items() if not c[5]]
PATH_FLIP_CASES = {"heston_qe", "heston_highpsi_qe"}          # hard (unsmoothed) QE branch indicators inside the path map


def _check_against_golden(sc, res, g, name):
    ours = sc.last_state["paths"].permute(2, 0, 1).cpu().numpy()
    assert np.allclose(ours, g["paths_main"], rtol=1e-11, atol=1e-13), np.abs(ours - g["paths_main"]).max()
    if "paths_pre" in g.files:
        pre = sc.last_state["paths_pre"].permute(2, 0, 1).cpu().numpy()
        assert np.allclose(pre, g["paths_pre"], rtol=1e-11, atol=1e-13)
    for i in range(len(sc.products)):
        key = f"expo_coeffs_{i}"
        if key in g.files and g[key].size:
            ref_c, our_c = g[key], sc.regression_coeffs[i].numpy()
            scale = np.maximum(np.abs(ref_c).max(), 1e-300)
            assert np.allclose(our_c, ref_c, rtol=1e-6, atol=1e-8 * scale), (name, i, np.abs(our_c - ref_c).max())
    n_ns = len(sc.netting_sets)
    if sc.last_state["cfs"] is not None:
        for ns_i in range(n_ns):
            ref = sum(g[f"cfs_{i}"] for i in range(len(sc.products)) if sc.product_to_netting_set_idx[i] == ns_i and f"cfs_{i}" in g.files)
            assert np.allclose(sc.last_state["cfs"][ns_i].cpu().numpy(), ref, rtol=1e-10, atol=1e-12)
    if sc.last_state["expo"] is not None:
        for ns_i in range(n_ns):
            ref = sum(g[f"exposures_{i}"] for i in range(len(sc.products)) if sc.product_to_netting_set_idx[i] == ns_i)
            assert np.allclose(sc.last_state["expo"][ns_i].cpu().numpy(), ref, rtol=1e-8, atol=1e-10)
    for ns_i in range(n_ns):
        for m_i, metric in enumerate(sc.risk_metrics.metrics):
            ref = g[f"result_{ns_i}_{m_i}"]
            ours = np.array([[v, e] for v, e in res.results[ns_i][m_i]], dtype=np.float64)
            assert np.allclose(ours[:, 0], ref[:, 0], rtol=1e-8, atol=1e-10), (name, metric.get_name(), ours[:, 0], ref[:, 0])
            assert np.allclose(ours[:, 1], ref[:, 1], rtol=1e-6, atol=1e-12), (name, metric.get_name(), ours[:, 1], ref[:, 1])


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
@pytest.mark.parametrize("name", NON_AAD)
def test_inject_z_against_reference(name, fused, hip):
    sc, g = cases.make_controller(name, hip, fused=fused)
    res = sc.run_simulation()
    _check_against_golden(sc, res, g, name)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
@pytest.mark.parametrize("name", NON_AAD)
def test_philox_gpu_vs_oracle(name, fused, hip, oracle):
    sc_g, _ = cases.make_controller(name, hip, inject=False, fused=fused)
    sc_c, _ = cases.make_controller(name, oracle, inject=False, fused=False)
    rg, rc = sc_g.run_simulation(), sc_c.run_simulation()
    pg = sc_g.last_state["paths"].cpu().numpy()
    pc = sc_c.last_state["paths"].numpy()
    # Only a HARD indicator inside the step map can make a path differ beyond rounding: the two branch tests of the Heston QE
    # step without smoothing (heston.py:214-240) may flip on a 1-ulp difference of psi or u for a vanishing fraction of paths.
    # Every other model's paths (and exercise decisions never touch the paths) must agree entry by entry.
    bad = ~np.isclose(pg, pc, rtol=1e-10, atol=1e-12)
    allowed = 1e-4 if name in PATH_FLIP_CASES else 0.0
    assert bad.mean() <= allowed, (name, bad.mean())
    for ns_i in range(len(sc_g.netting_sets)):
        for m_i, metric in enumerate(sc_g.risk_metrics.metrics):
            a = np.array(rg.results[ns_i][m_i], dtype=np.float64)
            b = np.array(rc.results[ns_i][m_i], dtype=np.float64)
            assert np.allclose(a[:, 0], b[:, 0], rtol=1e-8, atol=1e-10), (name, metric.get_name(), a[:, 0], b[:, 0])
            if not metric.get_name().startswith("pfe"):
                assert np.allclose(a[:, 1], b[:, 1], rtol=1e-5, atol=1e-11), (name, metric.get_name(), a[:, 1], b[:, 1])


def test_philox_paths_through_the_qe_scheme(hip, oracle):
    """Heston QE paths on identical counters: the whole RNG pipeline (normals + the extra uniform) through a scheme; the
    integer stream itself is compared word for word in test_philox_device_*; the normals may differ by libm ulps."""
    from mcx.common.enums import SimulationScheme
    from mcx.engine.engine import MonteCarloEngine
    from mcx.models.heston import HestonModel
    n = 1 << 14
    model = HestonModel(0.0, 1.0, 0.0, 0.5, -0.3, 1.0, 0.04, 0.04)
    tl = np.array([0.0, 0.5, 1.0])
    out = {}
    for be in (hip, oracle):
        eng = MonteCarloEngine(tl, SimulationScheme.QE, model, n, 3, backend=be, path_offset=123456789012)
        out[be.name] = eng.generate_paths_native().cpu().numpy()
    assert np.allclose(out["hip"], out["oracle"], rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("smoothing", [False, True], ids=["hard", "fuzzy"])
def test_heston_qe_random_parameters_gpu_vs_oracle(smoothing, hip, oracle):
    """the QE step with its reciprocals taken in groups from one v_rcp_f64 of a product (mcx_device.h) against the oracle's IEEE
    divisions, over 24 random parameter sets that reach both branches, tiny and large variances, strong vol-of-vol and long steps:
    the products of denominators must neither overflow nor lose the reference's eps floors.  Hard branch: a 1-ulp difference of
    psi or u may flip a branch for a vanishing share of the paths; fuzzy branch: the smoothed scheme's variance goes negative
     and a path may come arbitrarily close to the pole of 1/(psi + eps) — the same allowance."""
    from mcx.common.enums import SimulationScheme
    from mcx.engine.engine import MonteCarloEngine
    from mcx.models.heston import HestonModel
    rng = np.random.default_rng(20261005)
    n, worst = 8192, 0.0
    for case in range(24):
        v0 = float(10.0 ** rng.uniform(-6.0, -0.5))
        theta = float(10.0 ** rng.uniform(-6.0, -0.5))
        kappa = float(10.0 ** rng.uniform(-2.0, 1.2))
        sigma = float(10.0 ** rng.uniform(-2.0, 0.5))
        rho = float(rng.uniform(-0.95, 0.95))
        rate = float(rng.uniform(-0.02, 0.08))
        horizon = float(10.0 ** rng.uniform(-1.0, 1.0))
        model = HestonModel(0.0, 100.0, rate, sigma, rho, kappa, theta, v0)
        model.perform_smoothing = smoothing
        tl = np.linspace(horizon / 4, horizon, 4)
        out = {}
        steps = int(rng.integers(1, 6))
        for be in (hip, oracle):
            eng = MonteCarloEngine(tl, SimulationScheme.QE, model, n, steps, backend=be, path_offset=1000003 * case)
            out[be.name] = eng.generate_paths_native().cpu().numpy()
        # without smoothing every state is finite; with it the reference takes the root of a negative 2 / psi for many of these
        # parameter sets and the path is NaN from there on (13 of the 24 sets on the oracle): the same entries must be NaN here
        if not smoothing:
            assert np.all(np.isfinite(out["hip"])) and np.all(np.isfinite(out["oracle"])), (case, v0, theta, kappa, sigma)
        bad = ~np.isclose(out["hip"], out["oracle"], rtol=1e-9, atol=1e-12, equal_nan=True)
        worst = max(worst, float(bad.mean()))
        assert bad.mean() <= 2e-3, (case, smoothing, bad.mean(), dict(v0=v0, theta=theta, kappa=kappa, sigma=sigma, rho=rho, horizon=horizon, steps=steps))
    assert worst <= 2e-3


def test_random_model_parameters_gpu_vs_oracle(hip, oracle):
    """paths of every model family and scheme on random parameter sets — strong / vanishing mean reversion, volatilities over three
    decades, correlations up to +-0.95, CIR++ starts a hair above zero, a four-asset BlackScholesMulti with a random correlation
    matrix inside a ModelConfig with credit, Schwartz two-factor on a random curve, Hull-White on a random forward curve — against
    the oracle on identical Philox counters.  None of these steps contains a hard
    indicator: the paths must agree entry by entry."""
    from mcx.common.enums import SimulationScheme as SS
    from mcx.engine.engine import MonteCarloEngine
    from mcx.models.black_scholes import BlackScholesModel
    from mcx.models.black_scholes_multi import BlackScholesMulti
    from mcx.models.cirpp import CIRPPModel
    from mcx.models.model_config import ModelConfig
    from mcx.models.vasicek import VasicekModel
    rng = np.random.default_rng(7)
    n = 4096

    def lu(lo, hi):
        return float(10.0 ** rng.uniform(np.log10(lo), np.log10(hi)))

    def cir():
        kappa, theta = lu(0.02, 3.0), lu(1e-3, 0.2)
        vol = float(rng.uniform(0.05, 0.95)) * np.sqrt(2 * kappa * theta)              # inside the Feller condition the constructor asserts
        return CIRPPModel(0.0, "cp", cases.HAZARDS, kappa=kappa, theta=theta, volatility=vol, y0=lu(1e-9, 0.05))

    def vas(asset_id=None):
        return VasicekModel(0.0, float(rng.uniform(-0.01, 0.08)), float(rng.uniform(-0.01, 0.1)), lu(1e-3, 5.0), lu(1e-4, 0.1), asset_id=asset_id)

    def multi():
        a = rng.normal(size=(4, 4))
        c = a @ a.T
        d = np.sqrt(np.diag(c))
        ids = [f"a{k}" for k in range(4)]
        corr = 0.5 * np.eye(4) + 0.5 * c / np.outer(d, d)                                # (well inside the positive-definite cone: a credit column is added)
        return BlackScholesMulti(0.0, float(rng.uniform(0.0, 0.06)), ids, [lu(10.0, 500.0) for _ in ids], [lu(0.02, 0.9) for _ in ids], corr)

    def s2f():
        from mcx.models.schwartz_two_factor import SchwartzTwoFactorModel
        ts = [0.0, 0.5, 1.0, 2.0, 5.0, 25.0]
        return SchwartzTwoFactorModel(0.0, ts, [lu(10.0, 90.0) for _ in ts], rate=float(rng.uniform(0.0, 0.06)), short_term_mean_reversion=lu(1e-3, 5.0),
                                      short_term_vol=lu(0.02, 0.8), long_term_drift=float(rng.uniform(-0.05, 0.05)), long_term_vol=lu(0.01, 0.4),
                                      rho=float(rng.uniform(-0.9, 0.9)))

    def hw():
        from mcx.models.hull_white import HullWhiteModel
        ts = np.linspace(0.0, 25.0, 26)
        f = 0.03 + 0.02 * np.sin(ts / float(rng.uniform(2.0, 9.0))) + float(rng.uniform(-0.01, 0.02))
        return HullWhiteModel(0.0, float(f[0]), list(f), list(np.gradient(f, ts)), lu(1e-2, 2.0), lu(1e-3, 0.05), curve_times=list(ts))

    builders = [
        ("s2f", s2f, (SS.ANALYTICAL, SS.EULER)),
        ("hull_white", hw, (SS.ANALYTICAL, SS.EULER)),
        ("bs", lambda: BlackScholesModel(0, lu(1.0, 1e3), float(rng.uniform(-0.02, 0.1)), lu(0.01, 1.5)), (SS.ANALYTICAL, SS.EULER)),
        ("vasicek", lambda: vas(), (SS.ANALYTICAL, SS.EULER)),
        ("cirpp", cir, (SS.EULER,)),
        ("vasicek+cirpp", lambda: ModelConfig([vas("ir"), cir()], inter_asset_correlation_matrix=np.array([float(rng.uniform(-0.95, 0.95))])), (SS.EULER,)),
        ("multi", multi, (SS.ANALYTICAL, SS.EULER)),
        ("multi+cirpp", lambda: ModelConfig([multi(), cir()], inter_asset_correlation_matrix=[rng.uniform(-0.15, 0.15, size=(4, 1))]), (SS.EULER,)),
    ]
    for name, build, schemes in builders:
        for rep in range(6):
            model = build()
            horizon = lu(0.05, 20.0)
            tl = np.linspace(0.0, horizon, 6) if rep % 2 else np.linspace(horizon / 5, horizon, 5)
            steps = int(rng.integers(1, 5))
            for scheme in schemes:
                out = {}
                for be in (hip, oracle):
                    eng = MonteCarloEngine(tl, scheme, model, n, steps, backend=be, path_offset=977 * rep)
                    out[be.name] = eng.generate_paths_native().cpu().numpy()
                assert np.all(np.isfinite(out["oracle"])), (name, rep, scheme)
                bad = ~np.isclose(out["hip"], out["oracle"], rtol=1e-10, atol=1e-13)
                assert not bad.any(), (name, rep, scheme.name, float(bad.mean()), float(np.abs(out["hip"] - out["oracle"]).max()))


def _random_rate_book(case, metrics=None):
    """1-2 netting sets of 1-3 swaps / bonds / floaters with random schedules on Vasicek + CIR++ credit (the same book for a given case)"""
    from mcx.products.bond import Bond
    from mcx.products.swap import InterestRateSwap, IRSType
    r = np.random.default_rng(1000 + case)                       # the same book for both backends
    ir = cases.VasicekModel(0.0, float(r.uniform(0.0, 0.06)), float(r.uniform(0.0, 0.08)), float(10 ** r.uniform(-2, 0.5)),
                            float(10 ** r.uniform(-3, -1.3)), asset_id="ir")
    kappa, theta = float(10 ** r.uniform(-1.5, 0.3)), float(10 ** r.uniform(-2.5, -1))
    cr = cases.CIRPPModel(0.0, "cp", cases.HAZARDS, kappa=kappa, theta=theta, volatility=float(r.uniform(0.1, 0.9) * np.sqrt(2 * kappa * theta)),
                          y0=float(10 ** r.uniform(-5, -2)), deterministic=bool(r.integers(0, 4) == 0))
    model = cases.ModelConfig([ir, cr], inter_asset_correlation_matrix=np.array([float(r.uniform(-0.9, 0.9))]))
    sets, horizon = [], 0.0
    for k in range(int(r.integers(1, 3))):
        prods = []
        for q in range(int(r.integers(1, 4))):
            mat = float(r.choice([1.0, 1.5, 2.0, 3.0, 5.0]))
            horizon = max(horizon, mat)
            kind = int(r.integers(0, 3))
            if kind == 0:
                p = InterestRateSwap(0.0, mat, float(10 ** r.uniform(-1, 1.5)), float(r.uniform(0.0, 0.07)), float(r.choice([0.25, 0.5, 1.0])),
                                     float(r.choice([0.25, 0.5])), IRSType.PAYER if r.integers(0, 2) else IRSType.RECEIVER, "ir")
            elif kind == 1:
                p = Bond(0.0, mat, float(10 ** r.uniform(-1, 1)), float(r.choice([0.25, 0.5, 1.0])), True, float(r.uniform(0.0, 0.06)), "ir")
            else:
                p = Bond(0.0, mat, float(10 ** r.uniform(-1, 1)), float(r.choice([0.25, 0.5])), True, None, "ir")          # floater
            p.name = f"p{k}_{q}"
            prods.append(p)
        kw = {}
        if r.integers(0, 2):
            kw["threshold"] = float(10 ** r.uniform(-3, -1))
        if r.integers(0, 2):
            kw["margin_period_of_risk"] = float(r.choice([0.25, 0.5]))
        sets.append(cases.NettingSet(name=f"ns{k}", products=prods, counterparty_id="cp", **kw))
    tl = np.arange(0.0, horizon + 1e-9, 0.25) if r.integers(0, 2) else np.linspace(0.0, horizon, int(r.integers(5, 12)))
    mets = [cases.CVAMetric("cp", float(r.uniform(0.2, 0.6))), cases.PVMetric(), cases.EPEMetric(), cases.ENEMetric(), cases.PFEMetric(0.9)]
    return sets, model, cases.RiskMetrics(mets if metrics is None else metrics(), exposure_timeline=tl)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
def test_random_rate_books_gpu_vs_oracle(fused, hip, oracle):
    """the compiled date programs of the one-launch kernel and the event interpreter on RANDOM linear books: 1-4 swaps / bonds / floaters
    with random schedules on Vasicek + CIR++ credit, netting sets with random threshold and margin period of risk, exposure dates that
    do and do not coincide with payment dates, CVA + PV + EPE + ENE + PFE — every metric value against the oracle on identical counters"""
    rng = np.random.default_rng(11)
    for case in range(10):
        build = lambda: _random_rate_book(case)
        res = {}
        for be in (hip, oracle):
            ns, model, rm = build()
            sc = cases.SimulationController(ns, model, rm, 4096, 2048, int(rng.integers(1, 4)) if be is hip else steps, cases.E, backend=be)
            if be is hip:
                steps = sc.num_steps
                sc.allow_fused = fused
            res[be.name] = sc.run_simulation().results
        for ns_i in range(len(res["hip"])):
            for m_i in range(len(res["hip"][ns_i])):
                a, b = np.array(res["hip"][ns_i][m_i], dtype=np.float64), np.array(res["oracle"][ns_i][m_i], dtype=np.float64)
                assert np.allclose(a[:, 0], b[:, 0], rtol=1e-8, atol=1e-10), (case, ns_i, m_i, a[:, 0], b[:, 0])
                if m_i != 4:                                                    # (PFE carries no Monte-Carlo error)
                    assert np.allclose(a[:, 1], b[:, 1], rtol=1e-5, atol=1e-11), (case, ns_i, m_i, a[:, 1], b[:, 1])


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_equity_books_gpu_vs_oracle(seed, hip, oracle):
    """RANDOM books of every equity payoff family — Europeans, binaries, baskets (arithmetic / geometric), Asians, barriers of the four
    types, Americans, FlexiCalls — with random strikes, maturities, observation counts and weights on a four-asset BlackScholesMulti
    + CIR++ credit, collateralised or not: CVA + EPE + PV through the LSM regressions and the event interpreter, against the oracle"""
    from mcx.products.asian_option import AsianAveragingType, AsianOption
    from mcx.products.barrier_option import BarrierOption, BarrierOptionType
    from mcx.products.basket_option import BasketOption, BasketOptionType
    from mcx.products.bermudan_option import AmericanOption
    from mcx.products.binary_option import BinaryOption
    from mcx.products.flexicall import FlexiCall
    O, Eq = cases.OptionType, cases.Equity
    ids = [f"asset_{k}" for k in range(4)]

    def build():
        r = np.random.default_rng(500 + seed)
        corr = np.full((4, 4), float(r.uniform(0.0, 0.6))); np.fill_diagonal(corr, 1.0)
        spots = [float(r.uniform(60, 140)) for _ in ids]
        market = cases.BlackScholesMulti(0.0, float(r.uniform(0.0, 0.05)), ids, spots, [float(r.uniform(0.1, 0.5)) for _ in ids], corr)
        credit = cases.CIRPPModel(0.0, "cp", cases.HAZARDS, kappa=0.10, theta=0.01, volatility=0.02, y0=1e-4, deterministic=bool(seed % 2))
        model = cases.ModelConfig([market, credit], inter_asset_correlation_matrix=[np.full((4, 1), float(r.uniform(-0.2, 0.2)))])
        P = []
        typ = lambda: O.CALL if r.integers(0, 2) else O.PUT
        for i in range(24):
            k, a = int(r.integers(0, 7)), ids[int(r.integers(0, 4))]
            s0 = spots[ids.index(a)]
            mat, strike = float(r.choice([0.5, 0.75, 1.0, 1.5, 2.0])), float(s0 * r.uniform(0.8, 1.2))
            if k == 0:
                p = cases.EuropeanOption(Eq(a), mat, strike, typ(), asset_id=a)
            elif k == 1:
                p = BinaryOption(mat, strike, float(r.uniform(1, 20)), typ(), asset_id=a)
            elif k == 2:
                m = int(r.integers(2, 5))
                w = r.uniform(0.1, 1.0, size=m)
                p = BasketOption(mat, ids[:m], list(w / w.sum()), float(np.mean(spots[:m]) * r.uniform(0.85, 1.15)), typ(),
                                 BasketOptionType.ARITHMETIC if r.integers(0, 2) else BasketOptionType.GEOMETRIC, False)
            elif k == 3:
                p = AsianOption(0.0, mat, strike, int(r.integers(3, 13)), typ(),
                                AsianAveragingType.ARITHMETIC if r.integers(0, 2) else AsianAveragingType.GEOMETRIC, asset_id=a)
            elif k == 4:
                bt = [BarrierOptionType.UPANDOUT, BarrierOptionType.DOWNANDOUT, BarrierOptionType.UPANDIN, BarrierOptionType.DOWNANDIN][int(r.integers(0, 4))]
                up = bt in (BarrierOptionType.UPANDOUT, BarrierOptionType.UPANDIN)
                p = BarrierOption(0.0, mat, strike, int(r.integers(4, 13)), typ(), float(s0 * (r.uniform(1.1, 1.4) if up else r.uniform(0.6, 0.9))), bt, asset_id=a)
            elif k == 5:
                p = AmericanOption(Eq(a), mat, int(r.integers(3, 10)), strike, typ(), asset_id=a)
            else:
                L = int(r.integers(2, 5))
                und = [cases.EuropeanOption(Eq(a), float(t), float(s0 * r.uniform(0.9, 1.1)), O.CALL, asset_id=a) for t in np.linspace(mat / L, mat, L)]
                p = FlexiCall(und, int(r.integers(1, L)), asset_id=a)
            p.name = f"p{i}"
            P.append(p)
        horizon = max(float(p.modeling_timeline[-1]) for p in P)
        kw = dict(margin_period_of_risk=10 / 252) if seed % 2 else {}
        ns = cases.NettingSet(name="book", products=P, counterparty_id="cp", **kw)
        rm = cases.RiskMetrics([cases.CVAMetric("cp", 0.4), cases.EPEMetric(), cases.PVMetric()], exposure_timeline=np.linspace(0.0, horizon, 9))
        return [ns], model, rm

    out = {}
    for be in (hip, oracle):
        ns, model, rm = build()
        sc = cases.SimulationController(ns, model, rm, 768, 768, 1, cases.E, backend=be)
        res = sc.run_simulation()
        out[be.name] = [np.array(m, dtype=np.float64) for m in res.results[0]]
    for a, b in zip(out["hip"], out["oracle"]):
        assert np.allclose(a[:, 0], b[:, 0], rtol=1e-8, atol=1e-10), (seed, a[:, 0], b[:, 0])
        assert np.allclose(a[:, 1], b[:, 1], rtol=1e-6, atol=1e-10)


def _oracle_pairs(oracle, words):
    import ctypes as C
    n = words.shape[1]
    u, z = np.zeros((2, n)), np.zeros((2, n))
    fn = oracle.lib.orc_pair_from_words
    fn.restype = None
    a, b, c, d = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    for i in range(n):
        w = (C.c_uint32 * 4)(*[int(x) for x in words[:, i]])
        fn(w, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        u[:, i] = a.value, b.value
        z[:, i] = c.value, d.value
    return u, z


@pytest.mark.parametrize("bits", [7, 10])
def test_box_muller_edge_words_and_table_cells(bits, hip, oracle):
    """the word -> (uniform, normal pair) map of the kernels (mcx_box_muller) against the oracle's libm evaluation on the words
    that stress it: all ones (u rounds to 1: radius 0, no NaN from the unguarded root), all zeros (smallest u), the first / last
    word of table cells of both tables, exponent boundaries of the first uniform, and random words"""
    rng = np.random.default_rng(99)
    N = 1 << bits
    cols = [(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xfffff800, 0xffffffff, 0, 0), (0xfffff7ff, 0xffffffff, 1, 0),
            (0, 0, 0, 0), (0x800, 0, 0x7ff, 0), (0, 1, 0, 1), (0, 0x80000000, 0, 0x80000000), (0xffffffff, 0x7fffffff, 0xffffffff, 0x7fffffff)]
    for j in (0, 1, N // 2 - 1, N // 2, N - 2, N - 1):                       # trig table cell j: w3 top bits; log cells via u in [0.5, 1)
        lo_edge, hi_edge = j << (32 - bits), ((j + 1) << (32 - bits)) - 1
        cols += [(0, 0x80000000 | (lo_edge >> 1), 0, lo_edge), (0xffffffff, 0x80000000 | (hi_edge >> 1), 0xffffffff, hi_edge)]
    for e in range(1, 54, 4):                                               # first uniform ~ 2^-e
        hi = (1 << 32) >> e if e <= 32 else 0
        lo = 0 if e <= 32 else (1 << 64) >> e
        cols.append((lo & 0xffffffff, hi & 0xffffffff, 12345, 0x9e3779b9))
    words = np.array(cols, dtype=np.uint64).T
    words = np.concatenate([words, rng.integers(0, 1 << 32, size=(4, 4096), dtype=np.uint64)], axis=1).astype(np.uint32)
    u, z = hip.box_muller(torch.from_numpy(words.view(np.int32)).to(hip.device), bits)
    u, z = u.cpu().numpy(), z.cpu().numpy()
    uo, zo = _oracle_pairs(oracle, words)
    assert np.array_equal(u, uo)                                            # bit for bit, including u == 1.0 in column 0 and 1
    assert u[0, 0] == 1.0 and u[0, 1] == 1.0 and u[0, 2] < 1.0
    assert np.all(np.isfinite(z)) and z[0, 0] == 0.0 and z[1, 0] == 0.0 and z[0, 1] == 0.0
    assert np.abs(z - zo).max() < 2e-14, np.abs(z - zo).max()


KATS = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),          # Random123 kat_vectors, philox4x32 10
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox_device_words_match_random123_kats(hip):
    """the integer stream ON THE DEVICE, word for word (mcx_rng_draws; counter = (path_lo, path_hi, step, draw), key = seed)"""
    for ctr, key, exp in KATS:
        words, _, _ = hip.rng_draws(seed=key[0] | (key[1] << 32), path0=ctr[0] | (ctr[1] << 32), n=1, step=ctr[2], draw=ctr[3])
        got = tuple(int(w) & 0xffffffff for w in words.cpu().numpy()[:, 0])
        assert got == exp, (hex(got[0]), hex(exp[0]))


def test_philox_device_stream_bit_exact_vs_oracle(hip, oracle):
    """2^20 counters: device words == oracle words (array_equal), the 53-bit uniforms bit for bit, the table-driven Box-Muller
    pair against the oracle's libm pair to 1e-14 absolute"""
    import ctypes as C
    n, seed, path0, step, draw = 1 << 20, 43, (1 << 33) + 12345, 17, 1
    words, u, z = hip.rng_draws(seed, path0, n, step, draw)
    words = words.cpu().numpy().view(np.uint32)
    u, z = u.cpu().numpy(), z.cpu().numpy()
    # oracle words for every counter (vectorised restatement of orc_philox4x32_10 checked against the C oracle on a sample)
    path = np.uint64(path0) + np.arange(n, dtype=np.uint64)
    c = [(path & np.uint64(0xffffffff)).astype(np.uint64), (path >> np.uint64(32)).astype(np.uint64),
         np.full(n, step, np.uint64), np.full(n, draw, np.uint64)]
    k0, k1 = np.uint64(seed & 0xffffffff), np.uint64(seed >> 32)
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xffffffff)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & MASK, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & MASK]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    ref = np.stack(c).astype(np.uint32)
    for i in (0, 1, 777, n - 1):                                   # the numpy restatement IS the C oracle
        cc = (C.c_uint32 * 4)(int(path[i]) & 0xffffffff, int(path[i]) >> 32, step, draw)
        kk = (C.c_uint32 * 2)(seed & 0xffffffff, seed >> 32)
        oo = (C.c_uint32 * 4)()
        oracle.lib.orc_philox4x32_10(cc, kk, oo)
        assert tuple(oo) == tuple(int(x) for x in ref[:, i])
    assert np.array_equal(words, ref)
    x0 = (ref[1].astype(np.uint64) << np.uint64(32)) | ref[0].astype(np.uint64)
    x1 = (ref[3].astype(np.uint64) << np.uint64(32)) | ref[2].astype(np.uint64)
    u_ref = np.stack([((x0 >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53, ((x1 >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53])
    assert np.array_equal(u.view(np.uint64), u_ref.view(np.uint64))
    r = np.sqrt(-2.0 * np.log(u_ref[0]))
    z_ref = np.stack([r * np.cos(2.0 * np.pi * u_ref[1]), r * np.sin(2.0 * np.pi * u_ref[1])])
    assert np.abs(z - z_ref).max() < 1e-14 * max(1.0, np.abs(z_ref).max())


def test_lsm_mfma_matches_golden_coefficients(hip):
    """the MFMA Gram kernel against the reference's recorded regression coefficients (not only against the VALU kernel)"""
    sc, g = cases.make_controller("bermudan_swaption", hip)
    sc.use_mfma = True
    res = sc.run_simulation()
    _check_against_golden(sc, res, g, "bermudan_swaption[mfma]")


def test_lsm_mfma_matches_valu(hip):
    sc, g = cases.make_controller("bermudan_swaption", hip)
    sc.use_mfma = False
    sc.run_simulation()
    c0 = [c.clone() for c in sc.regression_coeffs]
    sc2, _ = cases.make_controller("bermudan_swaption", hip)
    sc2.use_mfma = True
    sc2.run_simulation()
    for a, b in zip(c0, sc2.regression_coeffs):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-12), (a - b).abs().max()


def test_radix_select_matches_sort(hip, oracle):
    from mcx.plan import UnsecuredSpec
    from mcx.parallel import Shard
    rng = np.random.default_rng(5)
    n, E = 20011, 5
    x = rng.standard_normal((E, n))
    x[1] = np.round(x[1], 1)          # heavy ties
    x[2] = 0.0                        # flat
    x[3, : n // 2] = -x[3, : n // 2] * 1e-300
    unsec = UnsecuredSpec(np.arange(E), None, 0.1, False)
    sc, _ = cases.make_controller("bs_european", hip, inject=False)
    q = 19000
    vals = sc._select_order_stats(Shard(), unsec, hip.from_numpy(x), [q - 1, q, q + 1])
    ref = oracle.pfe_sort(unsec, torch.from_numpy(x), q)
    assert np.array_equal(vals, ref)


@pytest.mark.parametrize("quantile", [0.95, 0.5, 0.999, 0.0005])
def test_bracket_select_matches_sort_and_the_digit_passes(quantile, hip):
    """PFE order statistics through ONE bracket pass + candidates (k5_bracket) against torch.sort and against the six digit passes:
    smooth rows, heavy ties, a flat row (every candidate buffer overflows: per-date fall-back), a row whose mass sits in two far
    clusters (the sample bracket may miss: per-date fall-back), threshold applied (netting_set.py:156-184)"""
    from mcx.plan import UnsecuredSpec
    from mcx.parallel import Shard
    rng = np.random.default_rng(11)
    n, E = 1 << 20, 6
    x = rng.standard_normal((E, n))
    x[1] = np.round(x[1], 2)                                   # heavy ties
    x[2] = 0.25                                                # flat
    x[3] = np.where(rng.random(n) < quantile, -1e3 + x[3], 1e3 + x[3])      # the wanted rank sits at the jump between two clusters
    x[4] = np.exp(3.0 * x[4])                                  # heavy right tail
    x[5, : n // 3] = 0.0                                       # an atom at zero (exposures of paths that exercised)
    unsec = UnsecuredSpec(np.arange(E), None, 0.1, False)
    sc, _ = cases.make_controller("bs_european", hip, inject=False)
    sc.num_paths_mainsim = n
    xt = hip.from_numpy(x)
    q = min(max(int(math.ceil(quantile * n)) - 1, 1), n - 2)
    ranks = [q - 1, q, q + 1]
    vals = sc._select_order_stats(Shard(), unsec, xt, ranks)
    info = sc.last_select
    # (at q = 0.5 the threshold's atom at zero holds the median of three rows: their candidates overflow and they fall back too)
    assert info["bracket_dates"] >= (3 if quantile > 0.9 else 1)
    assert info["point_dates"] >= 1 or not (0.01 < quantile < 0.99)        # the flat row (an open-ended bracket is not a point: it falls back)
    assert info["bracket_dates"] + info["point_dates"] + info["fallback_dates"] == E
    sc.bracket_select = False
    digits = sc._select_order_stats(Shard(), unsec, xt, ranks)
    u = torch.where(xt > 0.1, xt - 0.1, torch.where(xt < -0.1, xt + 0.1, torch.zeros_like(xt)))     # dev_thr
    ref = torch.sort(u, dim=1).values[:, ranks].cpu().numpy()
    assert np.array_equal(vals, digits)
    assert np.array_equal(vals, ref)


def test_statistical_anchor_bs_call(hip):
    """closed form 31.9648 (european_option.py:88-105) within 4 MC standard errors at 1 M Philox paths"""
    build, *_ = cases.CASES["bs_european"]
    from mcx.controller.controller import SimulationController
    ns, model, rm = build()
    sc = SimulationController(ns, model, rm, 1 << 20, 0, 4, cases.A, backend=hip)
    res = sc.run_simulation()
    pv, err = res.results[0][0][0]
    exact = float(ns[0].products[0].compute_pv_analytically(model))
    assert abs(pv - exact) < 4 * err, (pv, exact, err)
    assert 0.02 < err < 0.05


def test_fused_pass_is_used_and_lean(hip):
    """config-3-like book: the fused kernel is selected and, without `materialize`, no path/exposure tensor is created"""
    sc, _ = cases.make_controller("irs_cva", hip, inject=False)
    sc.materialize = False
    sc.main_plan = "fused"
    res = sc.run_simulation()
    assert sc._fused is not None and sc.timings.get("fused")
    assert sc.last_state["paths"] is None and sc.last_state["expo"] is None
    sc2, _ = cases.make_controller("irs_cva", hip, inject=False, fused=False)
    res2 = sc2.run_simulation()
    a, b = res.results[0][0][0], res2.results[0][0][0]
    assert np.isclose(a[0], b[0], rtol=1e-10) and np.isclose(a[1], b[1], rtol=1e-8)


def test_unfusable_books_fall_back(hip):
    sc, _ = cases.make_controller("netting", hip)          # collateralised netting set + unequal swap tenors
    sc.run_simulation()
    assert sc._fused is None


AAD = [n for n, c in cases.CASES.items() if c[5] and n not in cases.DRAWS_FROM]         # tangent-kernel cases
LSM_AAD = [n for n in cases.DRAWS_FROM]                                               # sensitivities through the regression


@pytest.mark.parametrize("name", AAD)
def test_tangent_kernel_against_reference_autograd(name, hip):
    """dual-number tangent kernel (csrc/kt_tangent.hip) vs the reference's torch.autograd gradients, recorded draws"""
    sc, g = cases.make_controller(name, hip)
    res = sc.run_simulation()
    ours = np.array(res.results[0][0], dtype=np.float64)
    assert np.allclose(ours, g["result_0_0"], rtol=1e-10), (ours, g["result_0_0"])
    grads = np.array(res.derivatives[0][0][0], dtype=np.float64)
    assert np.allclose(grads, g["grad_0_0"][0], rtol=1e-7, atol=1e-9), (grads, g["grad_0_0"][0])


@pytest.mark.parametrize("name", AAD)
def test_tangent_kernel_vs_oracle_complex_step(name, hip, oracle):
    sc_g, _ = cases.make_controller(name, hip, inject=False)
    sc_c, _ = cases.make_controller(name, oracle, inject=False)
    rg, rc = sc_g.run_simulation(), sc_c.run_simulation()
    assert np.allclose(np.array(rg.results[0][0]), np.array(rc.results[0][0]), rtol=1e-9)
    assert np.allclose(np.array(rg.derivatives[0][0][0]), np.array(rc.derivatives[0][0][0]), rtol=1e-7, atol=1e-9)


def test_bs_delta_anchor(hip):
    """config 2: BS call PV + Delta at 1M paths x 250 exact steps; closed-form delta N(d1) = 0.8750 within 4 SE"""
    import math
    from mcx.controller.controller import SimulationController
    ns, model, rm = cases.bs_european()
    sc = SimulationController(ns, model, rm, 1 << 20, 0, 250, cases.A, differentiate=True, backend=hip)
    res = sc.run_simulation()
    pv, err = res.results[0][0][0]
    d = res.get_derivatives(0, "pv", evaluation_idx=0)
    S0, K, r, sig, T = 120.0, 100.0, 0.05, 0.2, 2.0
    d1 = (math.log(S0 / K) + (r + 0.5 * sig * sig) * T) / (sig * math.sqrt(T))
    delta = 0.5 * (1 + math.erf(d1 / math.sqrt(2)))
    vega = S0 * math.exp(-0.5 * d1 * d1) / math.sqrt(2 * math.pi) * math.sqrt(T)
    assert abs(pv - 31.96482) < 4 * err
    assert abs(d["spot"] - delta) < 2e-3, (d["spot"], delta)
    assert abs(d["volatility"] - vega) / vega < 2e-2, (d["volatility"], vega)


@pytest.mark.parametrize("name", LSM_AAD)
def test_lsm_sensitivities_against_reference_autograd(name, hip):
    from test_oracle_golden import check_lsm_sensitivities
    sc, g = cases.make_controller(name, hip)
    res = sc.run_simulation()
    check_lsm_sensitivities(sc, g, res)


def test_exercise_products_take_the_forward_mode_path(hip):
    """Bermudan swaption sensitivities in ONE forward-mode pass (kt_lsm_step / kt_eval: exercise decisions from the primal values,
    tangents through the taken branch only — the reference's tape has no gradient through `should_exercise`,
    bermudan_option.py:122-128) against the reference's autograd, and against the 2P replayed bump runs they replace"""
    from test_oracle_golden import check_lsm_sensitivities
    sc, g = cases.make_controller("bermudan_swaption_aad", hip)
    res = sc.run_simulation()
    assert sc.timings.get("tangent") and sc.timings["forward_mode_passes"] == 1, sc.timings
    check_lsm_sensitivities(sc, g, res)
    sb, _ = cases.make_controller("bermudan_swaption_aad", hip)
    sb.forward_mode = False
    rb = sb.run_simulation()
    assert not sb.timings.get("tangent")
    for ns_i in range(len(res.derivatives)):
        for m_i in range(len(res.derivatives[ns_i])):
            a = np.array(res.derivatives[ns_i][m_i], dtype=np.float64)
            b = np.array(rb.derivatives[ns_i][m_i], dtype=np.float64)
            scale = np.abs(b).max(axis=1, keepdims=True) + 1e-12
            assert np.all(np.abs(a - b) <= 2e-6 * scale + 1e-9), (ns_i, m_i, a, b)


def _random_exercise_book(case):
    """a random Bermudan swaption on Vasicek (even cases) or American option on Black-Scholes (odd cases) with its exposure timeline"""
    from mcx.products.swap import InterestRateSwap, IRSType
    r = np.random.default_rng(300 + case)
    if case % 2 == 0:
        model = cases.VasicekModel(0.0, float(r.uniform(0.01, 0.05)), float(r.uniform(0.01, 0.07)), float(10 ** r.uniform(-1.5, 0)), float(10 ** r.uniform(-2.5, -1.6)))
        mat = float(r.choice([2.0, 3.0, 4.0]))
        und = InterestRateSwap(0.0, mat, 1.0, float(r.uniform(0.02, 0.05)), 0.25, 0.25, IRSType.PAYER if r.integers(0, 2) else IRSType.RECEIVER)
        n_ex = int(r.integers(4, 12))
        prod = cases.BermudanOption(und, [float(t) for t in np.linspace(mat / (n_ex + 1), mat * n_ex / (n_ex + 1), n_ex)], 0.0, cases.OptionType.CALL)
        tl = np.linspace(0.0, mat, 9)
    else:
        model = cases.BlackScholesModel(0.0, float(r.uniform(80, 120)), float(r.uniform(0.0, 0.06)), float(r.uniform(0.15, 0.6)))
        mat = float(r.choice([1.0, 2.0, 3.0]))
        prod = cases.AmericanOption(cases.Equity("id"), mat, int(r.integers(4, 13)), float(r.uniform(85, 115)),
                                    cases.OptionType.PUT if r.integers(0, 2) else cases.OptionType.CALL)
        tl = np.linspace(0.0, mat, 7)
    return prod, model, tl


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
@pytest.mark.parametrize("case", [0, 1, 2, 3, 4, 5])
def test_random_exercise_products_gpu_vs_oracle(case, fused, hip, oracle):
    """random Bermudan swaptions / American options through the LSM, the exercise date programs of the one-launch kernel (or the event
    interpreter) and the order-statistics select: EPE, PFE(0.95), ENE and PV against the oracle on identical counters.  An exercise
    decision is a hard comparison of two numbers that agree to ~1e-15: at 16,384 paths no path should sit that close to its boundary."""
    out = {}
    for be in (hip, oracle):
        prod, model, tl = _random_exercise_book(case)
        rm = cases.RiskMetrics([cases.EPEMetric(), cases.PFEMetric(0.95), cases.ENEMetric(), cases.PVMetric()], exposure_timeline=tl)
        sc = cases.SimulationController([cases.NettingSet(name="ex", products=[prod])], model, rm, 16384, 8192, 2, cases.E, backend=be)
        if be is hip:
            sc.allow_fused = fused
        out[be.name] = [np.array(m, dtype=np.float64) for m in sc.run_simulation().results[0]]
    for m_i, (a, b) in enumerate(zip(out["hip"], out["oracle"])):
        assert np.allclose(a[:, 0], b[:, 0], rtol=1e-8, atol=1e-10), (case, m_i, a[:, 0], b[:, 0])


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_forward_mode_through_random_exercise_products_against_replayed_bumps(case, hip):
    """random Bermudan swaptions (Vasicek) and American options (Black-Scholes): every EPE / ENE / PV sensitivity of the ONE forward-mode
    pass (frozen exercise policy, tangents through the taken branch) against central differences of bumped runs that REPLAY the base
    run's decisions — the derivative the reference's tape defines (bermudan_option.py:122-128)"""
    import mcx.aad as aad
    from mcx.helpers.host_threads import single_threaded_host

    def build():
        prod, model, tl = _random_exercise_book(case)
        mets = [cases.EPEMetric(), cases.ENEMetric(), cases.PVMetric()]
        return [cases.NettingSet(name="ex", products=[prod])], model, cases.RiskMetrics(mets, exposure_timeline=tl)

    grads = {}
    for tag, h in (("tangent", None), ("fd_small", 1e-6), ("fd_default", 1e-5)):
        ns, model, rm = build()
        sc = cases.SimulationController(ns, model, rm, 8192, 8192, 2, cases.E, differentiate=True, backend=hip)
        if h is None:
            r = sc.run_simulation()
            assert sc.timings.get("tangent") is True and sc.timings.get("forward_mode_passes") == 1, sc.timings
        else:
            saved = aad.bump_size
            aad.bump_size = lambda theta, h=h: h * max(abs(theta), 1e-2)
            try:
                with single_threaded_host():
                    r = aad.run_with_bumps(sc)
            finally:
                aad.bump_size = saved
        grads[tag] = r.derivatives
    for m_i in range(3):
        a = np.array(grads["tangent"][0][m_i], dtype=np.float64)
        fds = [np.array(grads[t][0][m_i], dtype=np.float64) for t in ("fd_small", "fd_default")]
        scale = max(np.abs(f).max() for f in fds) + 1e-300
        ok = np.zeros(a.shape, dtype=bool)
        for f in fds:
            ok |= np.isclose(a, f, rtol=2e-5, atol=2e-6 * scale)
        assert ok.all(), (case, m_i, a[~ok], [f[~ok] for f in fds])


@pytest.mark.parametrize("case", [0, 1, 2])
def test_european_tangent_kernel_on_random_books_against_bumps(case, hip):
    """kt_bs (csrc/kt_tangent.hip: Black-Scholes paths in dual numbers, several options in several netting sets) on random spot / rate /
    volatility, strikes and maturities, analytical and Euler scheme: PV sensitivities against central differences on the same counters"""
    import mcx.aad as aad
    from mcx.helpers.host_threads import single_threaded_host

    def build():
        r = np.random.default_rng(40 + case)
        model = cases.BlackScholesModel(0.0, float(r.uniform(50, 200)), float(r.uniform(-0.01, 0.08)), float(r.uniform(0.1, 0.7)))
        sets = []
        for k in range(int(r.integers(1, 4))):
            prods = []
            for q in range(int(r.integers(1, 4))):
                o = cases.EuropeanOption(cases.Equity(), float(r.choice([0.5, 1.0, 2.0, 3.0])), float(model.get_model_params()[0]) * float(r.uniform(0.7, 1.3)),
                                         cases.OptionType.CALL if r.integers(0, 2) else cases.OptionType.PUT)
                o.name = f"o{k}_{q}"
                prods.append(o)
            sets.append(cases.NettingSet(name=f"ns{k}", products=prods))
        return sets, model, cases.RiskMetrics([cases.PVMetric()]), (cases.A if case % 2 == 0 else cases.E)

    grads = {}
    for tag, h in (("tangent", None), ("fd", 1e-6)):
        ns, model, rm, scheme = build()
        sc = cases.SimulationController(ns, model, rm, 1 << 17, 0, 7, scheme, differentiate=True, backend=hip)
        if h is None:
            r = sc.run_simulation()
            assert sc.timings.get("tangent") is True, sc.timings
        else:
            saved = aad.bump_size
            aad.bump_size = lambda theta, h=h: h * max(abs(theta), 1e-2)
            try:
                with single_threaded_host():
                    r = aad.run_with_bumps(sc)
            finally:
                aad.bump_size = saved
        grads[tag] = r.derivatives
    for ns_i in range(len(grads["tangent"])):
        a, b = np.array(grads["tangent"][ns_i][0], dtype=np.float64), np.array(grads["fd"][ns_i][0], dtype=np.float64)
        assert np.allclose(a, b, rtol=2e-5, atol=2e-6 * np.abs(b).max()), (case, ns_i, a, b)


def test_basket_anchors_of_the_reference_tests(hip):
    """tests/pytests/test_model_config.py:18-71 and test_pv_basket_option.py:16-69 at their own sizes: arithmetic basket 12.60,
    geometric basket = its closed form 10.9551100513373 (ModelConfig of 4 BS models, 1 M paths; BlackScholesMulti with the
    geometric control variate)"""
    from mcx.models.black_scholes_multi import BlackScholesMulti
    from mcx.products.basket_option import BasketOption, BasketOptionType
    ids = ["asset1", "asset2", "asset3", "asset4"]

    def book(cv):
        b = BasketOption(1.0, ids, [0.25] * 4, 100, cases.OptionType.CALL, BasketOptionType.ARITHMETIC, cv); b.name = "basket_arithmetic"
        g = BasketOption(1.0, ids, [0.25] * 4, 100, cases.OptionType.CALL, BasketOptionType.GEOMETRIC); g.name = "basket_geometric"
        return b, g, [cases.NettingSet(name=b.get_name(), products=[b]), cases.NettingSet(name=g.get_name(), products=[g])]

    for scheme, steps in ((cases.A, 1), (cases.E, 50)):
        models = [cases.BlackScholesModel(0.0, 100.0, 0.0, 0.4, asset_id=a) for a in ids]
        model = cases.ModelConfig(models=models, inter_asset_correlation_matrix=np.array([[0.5]] * 6))
        b, g, ns = book(False)
        sc = cases.SimulationController(ns, model, cases.RiskMetrics([cases.PVMetric()]), 1000000, 0, steps, scheme, backend=hip)
        res = sc.run_simulation()
        (pa, ea), (pg, eg) = res.results[0][0][0], res.results[1][0][0]
        assert abs(pa - 12.60) < 0.02 + 3 * ea and abs(pg - 10.9551100513373) < 3.5 * eg + (0.02 if scheme == cases.E else 0.0), (scheme, pa, pg)
        assert abs(res.get_results(b.get_name(), "pv", evaluation_idx=0) - pa) == 0.0
    corr = np.full((4, 4), 0.5); np.fill_diagonal(corr, 1.0)
    model = BlackScholesMulti(0.0, 0.0, ids, [100.0] * 4, [0.4] * 4, corr)
    b, g, ns = book(True)
    assert abs(float(g.compute_pv_analytically(model)) - 10.9551100513373) < 1e-10
    sc = cases.SimulationController(ns, model, cases.RiskMetrics([cases.PVMetric()]), 1000000, 0, 1, cases.A, backend=hip)
    res = sc.run_simulation()
    (pa, ea), (pg, eg) = res.results[0][0][0], res.results[1][0][0]
    assert ea < 0.004 and abs(pa - 12.60) < 0.02 and abs(pg - 10.9551100513373) < 0.02 + 3 * eg, (pa, ea, pg, eg)


def test_large_book_product_batched_kernels_match_oracle(hip, oracle):
    """100 products of every payoff family on 384 paths: the product-batched LSM step (mcx_lsm_step_batch) and the
    product-chunked book kernel (mcx_eval_book with blockIdx.y = product chunk) against the per-product oracle, same Philox stream"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("large_book_tool", os.path.join(os.path.dirname(__file__), "..", "tools", "large_book.py"))
    lb = importlib.util.module_from_spec(spec); spec.loader.exec_module(lb)
    ids = [f"asset_{k}" for k in range(4)]
    out = {}
    for name, be in (("hip", hip), ("oracle", oracle)):
        corr = np.full((4, 4), 0.35); np.fill_diagonal(corr, 1.0)
        market = lb.BlackScholesMulti(0.0, 0.03, ids, [95.0 + 7.5 * k for k in range(4)], [0.18 + 0.03 * k for k in range(4)], corr)
        credit = lb.CIRPPModel(0.0, lb.CP, lb.HAZARDS, kappa=0.10, theta=0.01, volatility=0.02, y0=0.0001, deterministic=False)
        model = lb.ModelConfig([market, credit], inter_asset_correlation_matrix=[np.full((4, 1), 0.2)])
        products = lb.build_mixed_book(ids, 60, 6, 6, 8, 10, 6, 4)
        horizon = max(float(p.modeling_timeline[-1]) for p in products)
        ns = lb.NettingSet(name="book", products=products, counterparty_id=lb.CP, margin_period_of_risk=10 / 252)
        rm = lb.RiskMetrics([lb.CVAMetric(lb.CP, 0.4), cases.EPEMetric()], exposure_timeline=np.linspace(0.0, horizon, 16))
        sc = lb.SimulationController([ns], model, rm, 384, 384, 1, cases.E, backend=be)
        res = sc.run_simulation()
        out[name] = (np.array(res.results[0][0]), np.array(res.results[0][1]), sc.last_state["expo"].cpu().numpy())
    assert np.allclose(out["hip"][0], out["oracle"][0], rtol=1e-9, atol=1e-12), (out["hip"][0], out["oracle"][0])
    assert np.allclose(out["hip"][1], out["oracle"][1], rtol=1e-9, atol=1e-10)
    assert np.allclose(out["hip"][2], out["oracle"][2], rtol=1e-9, atol=1e-9)
    # the same book with the host solver behind the batched steps (the fallback a singular system takes): same coefficients as the
    # device solves of the run above
    dev = [c.numpy().copy() for c in sc.regression_coeffs]
    # ... and with the (step, solve) pairs enqueued one by one from Python (the several-GPU branch) instead of by mcx_lsm_run_batch:
    # the same kernels on the same tables, bit for bit
    sc.lsm_one_call = False
    sc._compiled_key = None
    sc.run_simulation()
    for a, b in zip(dev, [c.numpy() for c in sc.regression_coeffs]):
        assert np.array_equal(a, b)
    sc._lsm_host_solves = True
    sc._compiled_key = None
    sc.run_simulation()
    for a, b in zip(dev, [c.numpy() for c in sc.regression_coeffs]):
        assert np.allclose(a, b, rtol=1e-8, atol=1e-9 * max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize("book", ["ee", "pv"])
def test_large_book_exposure_and_pv_books_match_oracle(book, hip, oracle):
    """the other two of the reference's large-book scripts at 1/40 of their size (ee_performance_large_netting_set.py: EPE + PFE on an
    unsecured netting set, analytical scheme, Europeans through the closed-form exposure; pv_performance_large_netting_set.py:
    Monte-Carlo PV): HIP against the oracle on the same Philox stream"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("large_book_tool", os.path.join(os.path.dirname(__file__), "..", "tools", "large_book.py"))
    lb = importlib.util.module_from_spec(spec); spec.loader.exec_module(lb)
    ids = [f"asset_{k}" for k in range(4)]
    out = {}
    for name, be in (("hip", hip), ("oracle", oracle)):
        corr = np.full((4, 4), 0.35); np.fill_diagonal(corr, 1.0)
        market = lb.BlackScholesMulti(0.0, 0.03, ids, [95.0 + 7.5 * k for k in range(4)], [0.18 + 0.03 * k for k in range(4)], corr)
        products = lb.build_mixed_book(ids, 80, 4, 4, 6, 10, 6, 4)
        horizon = max(float(p.modeling_timeline[-1]) for p in products)
        ns = lb.NettingSet(name="book", products=products)
        if book == "ee":
            rm = lb.RiskMetrics([lb.EPEMetric(), lb.PFEMetric(0.95)], exposure_timeline=np.linspace(0.0, horizon, 12))
        else:
            rm = lb.RiskMetrics([lb.PVMetric()])
        sc = lb.SimulationController([ns], market, rm, 512, 512, 1, cases.A, backend=be)
        res = sc.run_simulation()
        out[name] = [np.array(m, dtype=np.float64) for m in res.results[0]]
    for a, b in zip(out["hip"], out["oracle"]):
        assert a.shape == b.shape and np.allclose(a[..., 0], b[..., 0], rtol=1e-9, atol=1e-9), (book, a, b)     # values
        assert np.allclose(a[..., 1], b[..., 1], rtol=1e-6, atol=1e-9)                                           # Monte-Carlo errors


def test_fused_kernel_timing_entry_points(hip):
    """mcx_fused_set_timing / mcx_fused_kernel_times: one duration per pass launched while armed, none when disarmed"""
    import bench
    sc = bench.build_controller(1 << 16, 4096, hip)
    sc.prepare()
    f = sc._fused
    assert hip.fused_kernel_times(f).size == 0
    hip.fused_set_timing(f, True)
    for _ in range(3):
        sc.main_pass()
    t = hip.fused_kernel_times(f)
    assert t.shape == (3,) and np.all(t > 0.0) and np.all(t < 50.0)
    assert hip.fused_kernel_times(f).size == 0                      # the ring re-arms empty
    hip.fused_set_timing(f, False)
    sc.main_pass()
    assert hip.fused_kernel_times(f).size == 0


def test_forward_mode_cva_against_reference_autograd_and_bumps(hip):
    """d CVA / d theta in dual numbers through pre-simulation, regression, book and CVA (csrc/kt_book.hip): vs the reference's
    torch.autograd gradients on its recorded draws, and vs common-random-number bumps in Philox mode (PV + CVA)"""
    from test_oracle_golden import check_lsm_sensitivities
    sc, g = cases.make_controller("irs_cva_aad", hip)
    res = sc.run_simulation()
    assert sc.timings.get("tangent") is True and sc.timings.get("forward_mode_passes") == 2, sc.timings
    check_lsm_sensitivities(sc, g, res)
    # mixed BS + Vasicek + deterministic CIR++ book, CVA and every EPE date, the unequal-tenor swap's per-term denominators
    sc, g = cases.make_controller("mixed_cva_aad", hip)
    res = sc.run_simulation()
    assert sc.timings.get("tangent") is True and sc.timings.get("forward_mode_passes") == 3, sc.timings
    check_lsm_sensitivities(sc, g, res)
    out = {}
    for fwd in (True, False):
        ns, model, _ = cases.irs_cva()
        rm = cases.RiskMetrics([cases.CVAMetric("cp", 0.4), cases.PVMetric(), cases.EEPEMetric(), cases.CEMetric(), cases.ENEMetric()],
                               exposure_timeline=np.arange(11) * 0.25)
        sc = cases.SimulationController(ns, model, rm, 8192, 4096, 3, cases.E, differentiate=True, backend=hip)
        sc.forward_mode = fwd
        r = sc.run_simulation()
        assert bool(sc.timings.get("tangent")) == fwd
        out[fwd] = (np.array(r.derivatives[0][0][0]), np.array(r.derivatives[0][1][0]), np.array(r.results[0][0]),
                    np.array(r.derivatives[0][2][0]), np.array(r.derivatives[0][3][0]), np.array(r.derivatives[0][4][3]))
    for k in (0, 1, 3, 4, 5):
        scale = np.abs(out[False][k]).max()
        assert np.allclose(out[True][k], out[False][k], rtol=2e-5, atol=2e-6 * scale), (k, out[True][k], out[False][k])
    assert np.allclose(out[True][2], out[False][2], rtol=1e-12)


@pytest.mark.parametrize("case", [0, 3, 5, 8])
def test_forward_mode_on_random_rate_books_against_bumps(case, hip):
    """the dual-number pass (csrc/kt_book.hip: paths, regressions, book, CVA / EPE / ENE / PV in dual numbers) on random linear books
    against central differences on the same Philox counters: every sensitivity of every metric date"""
    import mcx.aad as aad
    from mcx.helpers.host_threads import single_threaded_host
    mets = lambda: [cases.CVAMetric("cp", 0.4), cases.PVMetric(), cases.EPEMetric(), cases.ENEMetric()]
    # Two step sizes: thresholds and margin periods put kinks into the exposures, and a central difference over a kink is off by
    # O(h) (case 5: d CVA / d sigma_r is 7.22328124e-3 in dual numbers and at h = 1e-6 theta, 7.22246e-3 at 1e-5 theta), while the
    # small parameters of the credit model lose digits to cancellation at the small step.  Every component must agree with one of them.
    grads = {}
    for tag, h in (("tangent", None), ("fd_small", 1e-6), ("fd_default", 1e-5)):
        ns, model, rm = _random_rate_book(case, mets)
        sc = cases.SimulationController(ns, model, rm, 8192, 4096, 2, cases.E, differentiate=True, backend=hip)
        if h is None:
            r = sc.run_simulation()
            assert sc.timings.get("tangent") is True, sc.timings
        else:
            saved = aad.bump_size
            aad.bump_size = lambda theta, h=h: h * max(abs(theta), 1e-2)
            try:
                with single_threaded_host():
                    r = aad.run_with_bumps(sc)
            finally:
                aad.bump_size = saved
        grads[tag] = r.derivatives
    for ns_i in range(len(grads["tangent"])):
        for m_i in range(len(grads["tangent"][ns_i])):
            a = np.array(grads["tangent"][ns_i][m_i], dtype=np.float64)
            fds = [np.array(grads[t][ns_i][m_i], dtype=np.float64) for t in ("fd_small", "fd_default")]
            scale = max(np.abs(f).max() for f in fds) + 1e-300
            ok = np.zeros(a.shape, dtype=bool)
            for f in fds:
                ok |= np.isclose(a, f, rtol=2e-5, atol=2e-6 * scale)
            assert ok.all(), (case, ns_i, m_i, a[~ok], [f[~ok] for f in fds])


def test_table_box_muller_normals_moments_and_tails(hip):
    """the table-driven Box-Muller of the path kernels (mcx_math.h): both outputs of a draw (cos and sin branch) over 2 x 2^24
    samples — mean, variance, skewness, kurtosis, cross-correlation and the 4-sigma / 5-sigma tail masses of N(0,1)"""
    n = 1 << 24
    models = [cases.BlackScholesModel(0.0, 1.0, 0.0, 1.0, asset_id=a) for a in ("a", "b")]
    model = cases.ModelConfig(models=models, inter_asset_correlation_matrix=np.array([[0.0]]))
    eng = cases.MonteCarloEngine(np.array([1.0]), cases.A, model, n, 1, backend=hip) if hasattr(cases, "MonteCarloEngine") else None
    if eng is None:
        from mcx.engine.engine import MonteCarloEngine
        eng = MonteCarloEngine(np.array([1.0]), cases.A, model, n, 1, backend=hip)
    p = eng.generate_paths_native()                       # [1][2][n]: S_T = exp(-0.5 + z)
    z = torch.log(p[0]) + 0.5
    for k in range(2):
        x = z[k]
        m, v = float(x.mean()), float(x.var())
        s = float(((x - m) ** 3).mean()) / v ** 1.5
        kurt = float(((x - m) ** 4).mean()) / v ** 2
        assert abs(m) < 5 / np.sqrt(n) and abs(v - 1) < 5 * np.sqrt(2 / n), (k, m, v)
        assert abs(s) < 5 * np.sqrt(6 / n) and abs(kurt - 3) < 5 * np.sqrt(24 / n), (k, s, kurt)
        for thr, prob in ((4.0, 6.334248366623973e-05), (5.0, 5.733031437583878e-07)):
            cnt, exp = int((x.abs() > thr).sum()), prob * n
            assert abs(cnt - exp) < 5 * np.sqrt(exp) + 3, (k, thr, cnt, exp)
    assert abs(float((z[0] * z[1]).mean())) < 5 / np.sqrt(n)


def test_rccl_collectives_single_rank_group(hip):
    """the collectives the N > 1 path uses (mcx/parallel.py: all_gather_into_tensor of accumulator records, all_reduce of LSM
    moments / select histograms, broadcast) on a one-rank RCCL group: proves the backend initialises on this stack; the
    multi-rank logic itself is covered by the 2-rank gloo tests"""
    import os
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from mcx.parallel import Shard
        sh = Shard()
        assert sh.active and sh.backend == "nccl" and sh.split(10) == (0, 10)
        rec = torch.arange(8, dtype=torch.float64, device="cuda").reshape(2, 4)
        out = torch.empty((1, 2, 4), dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(out, rec)
        assert torch.equal(out[0], rec)
        t = torch.ones(5, dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        dist.broadcast(t, 0)
        assert float(t.sum()) == 5.0
        sc, _ = cases.make_controller("irs_cva", hip, inject=False)       # a whole run with an active (one-rank) process group
        a = sc.run_simulation().results[0][0][0]
        # two passes in flight (gather + host copy of pass k behind the kernel of pass k+1): same records as the plain pass
        sc.materialize = False
        assert sc.pipelined_passes_available()
        t1 = sc.fused_pass_begin(); t2 = sc.fused_pass_begin()
        r1 = sc.fused_pass_end(t1); t3 = sc.fused_pass_begin(); r2 = sc.fused_pass_end(t2); r3 = sc.fused_pass_end(t3)
        plain = tuple(sc._fused_pass()[0][0][0])                  # the same pass without the pipeline: identical records
        assert tuple(r1[0][0][0]) == plain and tuple(r2[0][0][0]) == plain and tuple(r3[0][0][0]) == plain
        assert np.allclose(plain, a, rtol=1e-12)                  # (the materialising run takes the unmerged CVA date path)
        with pytest.raises(RuntimeError):
            ta, tb = sc.fused_pass_begin(), sc.fused_pass_begin()
            try:
                sc.fused_pass_begin()
            finally:
                sc.fused_pass_end(ta); sc.fused_pass_end(tb)
        # Bermudan swaption with EPE + PFE: all-reduce of the LSM moments per date and of the select histograms per digit pass
        sb, _ = cases.make_controller("bermudan_swaption", hip, inject=False)
        ab = sb.run_simulation().results
    finally:
        dist.destroy_process_group()
    sc2, _ = cases.make_controller("irs_cva", hip, inject=False)
    b = sc2.run_simulation().results[0][0][0]
    assert a[0] == b[0]
    sc2.materialize = False
    assert sc2.pipelined_passes_available()                  # without a process group: the same pipeline, no collective
    assert tuple(sc2.fused_pass_end(sc2.fused_pass_begin())[0][0][0]) == tuple(sc2._fused_pass()[0][0][0])
    sb2, _ = cases.make_controller("bermudan_swaption", hip, inject=False)
    bb = sb2.run_simulation().results
    for m in range(len(bb[0])):
        np.testing.assert_allclose(np.array(ab[0][m], dtype=float), np.array(bb[0][m], dtype=float), rtol=1e-9, atol=1e-12)


def test_simulate_time_step_api_on_the_gpu(hip):
    """the reference's Model.simulate_time_step_* signature served by mcx_generate_paths_from_state (per-path start states)"""
    from test_oracle_units import check_simulate_time_step_api
    check_simulate_time_step_api(hip)


def test_comm_entry_points_single_rank(hip):
    """mcx_comm_* of the C ABI (RCCL loaded with dlopen): a one-rank communicator, all-reduce and all-gather of device buffers"""
    uid = hip.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    hip.comm_init(1, 0, uid)
    try:
        t = torch.arange(1000, dtype=torch.float64, device=hip.device) * 0.5
        ref = t.clone()
        hip.allreduce_(t)
        g = hip.allgather(ref, 1)
        torch.cuda.synchronize()
        assert torch.equal(t, ref) and g.shape == (1, 1000) and torch.equal(g[0], ref)
    finally:
        hip.comm_destroy()


def test_forward_mode_collateralised_netting_set_matches_bumps(hip):
    """margin-period-of-risk collateral and a threshold in the dual CVA / profile kernels (netting_set.py:110-184): forward mode
    vs common-random-number bumps, same Philox stream"""
    from mcx.products.swap import InterestRateSwap, IRSType
    out = {}
    for fwd in (True, False):
        _, model, _ = cases.irs_cva()
        irs = InterestRateSwap(0.0, 2.5, 1.0, 0.03, 0.25, 0.25, IRSType.PAYER, "irs")
        ns = [cases.NettingSet(name="coll", products=[irs], counterparty_id="cp", threshold=0.002, margin_period_of_risk=0.25)]
        rm = cases.RiskMetrics([cases.CVAMetric("cp", 0.4), cases.EPEMetric()], exposure_timeline=np.arange(11) * 0.25)
        sc = cases.SimulationController(ns, model, rm, 8192, 4096, 3, cases.E, differentiate=True, backend=hip)
        sc.forward_mode = fwd
        r = sc.run_simulation()
        assert bool(sc.timings.get("tangent")) == fwd, sc.timings
        out[fwd] = (np.array(r.derivatives[0][0][0]), np.array(r.derivatives[0][1]), np.array(r.results[0][0]))
    assert np.allclose(out[True][2], out[False][2], rtol=1e-12)
    for k in (0, 1):
        scale = np.abs(out[False][k]).max()
        # (a bumped path crossing the threshold / relu kink shows up as ~1e-5 in ONE entry of the finite difference)
        assert scale > 0 and np.allclose(out[True][k], out[False][k], rtol=5e-5, atol=3e-5 * scale), (k, out[True][k], out[False][k])


def test_forward_mode_falls_back_to_bumps_for_events_without_tangent_form(hip):
    """a binary payoff has no dual form in kt_book.hip: the library answers MCX_E_NOT_FUSABLE and differentiate=True completes
    with common-random-number bumps"""
    model = cases.BlackScholesModel(0.0, 100.0, 0.03, 0.2, asset_id="asset")
    prod = cases.BinaryOption(1.0, 100.0, 10.0, cases.OptionType.CALL, asset_id="asset")
    sc = cases.SimulationController([cases.NettingSet(name="bin", products=[prod])], model, cases.RiskMetrics([cases.PVMetric()]),
                                    4096, 0, 4, cases.E, differentiate=True, backend=hip)
    res = sc.run_simulation()
    assert sc.timings.get("tangent") is False and sc.timings.get("bumped_passes") == 6
    d = res.get_derivatives("bin", "pv", evaluation_idx=0)
    assert all(np.isfinite(float(v)) for v in d.values()) and float(d["spot"]) > 0.0


def test_forward_mode_analytic_black_scholes_exposures_match_bumps(hip):
    """EPE / PV sensitivities of a European option book whose exposures are the Black-Scholes closed form
    (european_option.py:123-145, MCX_EV_EXPO_BS): dual-number evaluation vs common-random-number bumps"""
    out = {}
    for fwd in (True, False):
        model = cases.BlackScholesModel(0.0, 100.0, 0.03, 0.25)
        c = cases.EuropeanOption(cases.Equity(), 1.0, 95.0, cases.OptionType.CALL); c.name = "call"
        p = cases.EuropeanOption(cases.Equity(), 1.5, 110.0, cases.OptionType.PUT); p.name = "put"
        ns = [cases.NettingSet(name="opts", products=[c, p])]
        rm = cases.RiskMetrics([cases.EPEMetric(), cases.PVMetric(), cases.PFEMetric(0.95)],
                               exposure_timeline=np.array([0.0, 0.25, 0.5, 1.0, 1.25, 1.5, 2.0]))
        sc = cases.SimulationController(ns, model, rm, 16384, 0, 4, cases.E, differentiate=True, backend=hip)
        sc.forward_mode = fwd
        r = sc.run_simulation()
        assert bool(sc.timings.get("tangent")) == fwd, sc.timings
        out[fwd] = (np.array(r.derivatives[0][0]), np.array(r.derivatives[0][1]), np.array(r.results[0][0]), np.array(r.derivatives[0][2]))
    assert np.allclose(out[True][2], out[False][2], rtol=1e-12)
    for k in (0, 1):
        scale = np.abs(out[False][k]).max()
        assert scale > 0 and np.allclose(out[True][k], out[False][k], rtol=1e-4, atol=3e-5 * scale), (k, out[True][k], out[False][k])
    # PFE(0.95): the tangent of the path at the order statistic; a bump keeps the same path there unless two paths swap ranks
    # (when a bump makes two paths next to the quantile swap ranks the difference quotient of that entry is not the tangent of
    #  one path any more: require agreement of at least 90 % of the entries, exactly as many as never see a swap)
    scale = np.abs(out[False][3]).max()
    close = np.isclose(out[True][3], out[False][3], rtol=1e-4, atol=1e-6 * scale)
    assert scale > 0 and close.mean() >= 0.9, (out[True][3], out[False][3])
