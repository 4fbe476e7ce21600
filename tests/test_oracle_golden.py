"""Pins the CPU oracle (oracle/mcx_oracle.c) AND the product's host logic (descriptor compilation, LSM solve, metric
finalisation, radix-select driver) to the reference: the recorded torch draws are replayed through the oracle and every
tensor the reference produced must come back."""
import numpy as np
import pytest
import torch

import cases

NON_AAD = [n for n, c in cases.CASES.items() if not c[5]]


def _rel(a, b, floor=1e-300):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


@pytest.mark.parametrize("name", NON_AAD)
def test_case_against_reference(name, oracle):
    sc, g = cases.make_controller(name, oracle)
    res = sc.run_simulation()
    # timelines (float-keyed: must be identical)
    assert np.array_equal(sc.simulation_timeline.numpy(), g["simulation_timeline"])
    assert np.array_equal(sc.exposure_timeline.numpy(), g["exposure_timeline"])
    # paths [N,T,D] in the reference; ours [T,D,N]
    ours = sc.last_state["paths"].permute(2, 0, 1).numpy()
    assert ours.shape == g["paths_main"].shape
    assert np.allclose(ours, g["paths_main"], rtol=1e-12, atol=1e-14), _rel(ours, g["paths_main"])
    if "paths_pre" in g.files:
        pre = sc.last_state["paths_pre"].permute(2, 0, 1).numpy()
        assert np.allclose(pre, g["paths_pre"], rtol=1e-12, atol=1e-14)
    # regression coefficients: fitted values agree (coefficients themselves are conditioning-limited in the reference)
    for i, p in enumerate(sc.products):
        key = f"expo_coeffs_{i}"
        if key in g.files and g[key].size:
            ref_c, our_c = g[key], sc.regression_coeffs[i].numpy()
            scale = np.maximum(np.abs(ref_c).max(), 1e-300)
            assert np.allclose(our_c, ref_c, rtol=1e-6, atol=1e-8 * scale), (name, i, np.abs(our_c - ref_c).max())
    # per-product cashflows / exposures summed per netting set
    n_ns = len(sc.netting_sets)
    if sc.last_state["cfs"] is not None:
        for ns_i in range(n_ns):
            ref = sum(g[f"cfs_{i}"] for i in range(len(sc.products)) if sc.product_to_netting_set_idx[i] == ns_i and f"cfs_{i}" in g.files)
            ours = sc.last_state["cfs"][ns_i].numpy()
            assert np.allclose(ours, ref, rtol=1e-10, atol=1e-12), (name, "cfs", np.abs(ours - ref).max())
    if sc.last_state["expo"] is not None:
        for ns_i in range(n_ns):
            ref = sum(g[f"exposures_{i}"] for i in range(len(sc.products)) if sc.product_to_netting_set_idx[i] == ns_i)
            ours = sc.last_state["expo"][ns_i].numpy()
            assert np.allclose(ours, ref, rtol=1e-8, atol=1e-10), (name, "expo", np.abs(ours - ref).max())
    # metric values and MC errors
    for ns_i in range(n_ns):
        for m_i, metric in enumerate(sc.risk_metrics.metrics):
            ref = g[f"result_{ns_i}_{m_i}"]
            ours = np.array([[v, e] for v, e in res.results[ns_i][m_i]], dtype=np.float64)
            assert ours.shape == ref.shape
            assert np.allclose(ours[:, 0], ref[:, 0], rtol=1e-8, atol=1e-10), (name, metric.get_name(), ours[:, 0], ref[:, 0])
            # MC errors: tiny "deterministic" errors (1e-19 in the reference) only need to be tiny here too
            assert np.allclose(ours[:, 1], ref[:, 1], rtol=1e-6, atol=1e-12), (name, metric.get_name(), ours[:, 1], ref[:, 1])
    assert list(res.metric_names) == list(g["metric_names"])
    assert list(res.netting_set_names) == list(g["netting_set_names"])


AAD = [n for n, c in cases.CASES.items() if c[5] and n not in cases.DRAWS_FROM]         # tangent-kernel cases
LSM_AAD = [n for n in cases.DRAWS_FROM]                                               # sensitivities through the regression


@pytest.mark.parametrize("name", AAD)
def test_tangents_against_reference_autograd(name, oracle):
    """differentiate=True: PV (smoothed primal) and d PV / d theta for every model parameter vs torch.autograd"""
    sc, g = cases.make_controller(name, oracle)
    res = sc.run_simulation()
    ref = g["result_0_0"]
    ours = np.array(res.results[0][0], dtype=np.float64)
    assert np.allclose(ours, ref, rtol=1e-10), (ours, ref)
    grads = np.array(res.derivatives[0][0][0], dtype=np.float64)
    assert np.allclose(grads, g["grad_0_0"][0], rtol=1e-7, atol=1e-9), (grads, g["grad_0_0"][0])
    assert list(res.model_param_names) == list(g["param_names"])
    d = res.get_derivatives(0, "pv", evaluation_idx=0)
    assert set(d) == set(g["param_names"])


def check_lsm_sensitivities(sc, g, res):
    """CVA / EPE sensitivities through the LSM regression (reference: torch.autograd through lstsq, controller.py:609-627) vs the
    common-random-number central differences of mcx.aad.run_with_bumps on the reference's recorded draws"""
    assert list(res.model_param_names) == list(g["param_names"])
    for ns_i in range(len(res.results)):
        for m_i in range(len(res.results[ns_i])):
            ref_v = g[f"result_{ns_i}_{m_i}"]
            ours_v = np.array(res.results[ns_i][m_i], dtype=np.float64)
            assert np.allclose(ours_v[:, 0], ref_v[:, 0], rtol=1e-8, atol=1e-10)
            ref = g[f"grad_{ns_i}_{m_i}"]
            ours = np.array(res.derivatives[ns_i][m_i], dtype=np.float64)
            assert ours.shape == ref.shape
            unused = np.isnan(ref)                   # parameters the reference's tape never touches (deterministic credit)
            scale = np.max(np.abs(ref[~unused].reshape(ref.shape[0], -1)), axis=1, keepdims=True) if (~unused).any() else 1.0
            # ... come back as zero up to one ulp of the value divided by the bump (2.8e-10 for CVA 0.29, h = 1e-7)
            assert np.all(np.abs(np.where(unused, ours, 0.0)) <= 1e-8 * np.maximum(np.broadcast_to(scale, ref.shape), 1e-3))
            err = np.abs(np.where(unused, 0.0, ours - ref))
            tol = 2e-6 * np.maximum(np.abs(np.where(unused, 0.0, ref)), 1e-3 * np.broadcast_to(scale, ref.shape)) + 1e-10
            assert np.all(err <= tol), (ns_i, m_i, ours, ref)


@pytest.mark.parametrize("name", LSM_AAD)
def test_lsm_sensitivities_against_reference_autograd(name, oracle):
    sc, g = cases.make_controller(name, oracle)
    res = sc.run_simulation()
    check_lsm_sensitivities(sc, g, res)


def test_netting_set_interpolated_collateral_matches_reference():
    """NettingSet.compute_collateral_profile / compute_unsecured_exposure_profiles without exact delayed indices
    (netting_set.py:76-107, 151-157): values recorded from the reference for 'linear' / 'previous', thresholds, three MPoRs"""
    g = cases.load_golden("netting_interp")
    tl, expo = torch.from_numpy(g["timeline"]), torch.from_numpy(g["expo"])
    prod = cases.EuropeanOption(cases.Equity("eq"), 1.0, 100.0, cases.OptionType.CALL)
    n = 0
    for key in [k[5:] for k in g.files if k.startswith("coll_")]:
        mode, thr, mpor = key.split("_")
        ns = cases.NettingSet(name="c", products=[prod], threshold=float(thr), margin_period_of_risk=float(mpor), collateral_interpolation=mode)
        assert np.allclose(ns.compute_collateral_profile(expo, tl).numpy(), g["coll_" + key], rtol=1e-14, atol=1e-14), key
        assert np.allclose(ns.compute_unsecured_exposure_profiles(expo, tl).numpy(), g["unsec_" + key], rtol=1e-14, atol=1e-14), key
        n += 1
    assert n == 12
