"""Unit-level pins of the oracle and the host logic against reference-generated fixtures (tests/golden/steps.npz,
paths_mc4.npz) and published known-answer vectors."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import cases
from mcx import _abi
from mcx.common.enums import SimulationScheme
from mcx.engine.engine import MonteCarloEngine
from mcx.models.black_scholes import BlackScholesModel
from mcx.models.cirpp import CIRPPModel
from mcx.models.heston import HestonModel
from mcx.models.model_config import ModelConfig
from mcx.models.vasicek import VasicekModel
from mcx.plan import SimPlan, UnsecuredSpec, solve_normal_equations

G = np.load(os.path.join(cases.GOLDEN, "steps.npz"))


def test_philox_known_answers(oracle):
    """Random123 kat_vectors for philox4x32 10 rounds"""
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kats:
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        oracle.lib.orc_philox4x32_10(c, k, o)
        assert tuple(o) == exp


def _one_step(oracle, model, scheme, state, z, u=None):
    """push recorded (state, z) through the oracle's sub-step [t1, t2] by making it the only step of a plan"""
    t1, t2 = float(G["t1"][0]), float(G["t2"][0])
    plan = SimPlan(model, np.array([t2]), scheme, 1)
    # overwrite the single sub-step with the fixture's interval and aux
    dt = t2 - t1
    plan.steps["dt"], plan.steps["sqrt_dt"], plan.steps["t1"] = dt, np.sqrt(dt), t1
    aux = model._step_aux(scheme, t1, dt)
    plan.aux[:] = 0
    for s, vals in enumerate(aux):
        plan.aux[0, s, :len(vals)] = vals
    plan.chol[0] = model.get_cholesky(scheme, dt).numpy()
    n = state.shape[0]
    out = torch.empty(1, plan.n_state, n, dtype=torch.float64)
    # the engine starts from the model's initial state; emulate an arbitrary start state by looping paths one at a time
    res = np.zeros((n, plan.n_state))
    zt = torch.from_numpy(np.ascontiguousarray(z.T[None]))             # [1][n_z][N]
    ut = torch.from_numpy(np.ascontiguousarray(u[None])) if u is not None else None
    sim = oracle.sim_create(plan)
    init_backup = plan.init_state.copy()
    # vectorised: oracle reads init_state once per path, so run N single-path simulations in chunks sharing the state
    for i in range(n):
        plan.init_state[:] = state[i]
        o = oracle.generate_paths(sim, 0, 0, 1, inject_z=zt[:, :, i:i + 1].contiguous(),
                                  inject_u=ut[:, i:i + 1].contiguous() if ut is not None else None)
        res[i] = o[0, :, 0].numpy()
    plan.init_state[:] = init_backup
    return res


def test_step_maps_match_reference(oracle):
    A, E, Q = SimulationScheme.ANALYTICAL, SimulationScheme.EULER, SimulationScheme.QE
    bs = BlackScholesModel(0.0, 120.0, 0.05, 0.2)
    assert np.allclose(_one_step(oracle, bs, A, G["bs_state"], G["bs_z"]), G["bs_exact"], rtol=1e-13)
    assert np.allclose(_one_step(oracle, bs, E, G["bs_state"], G["bs_z"]), G["bs_euler"], rtol=1e-13)
    va = VasicekModel(0.0, 0.03, 0.05, 0.1, 0.01, asset_id="irs")
    assert np.allclose(_one_step(oracle, va, A, G["va_state"], G["va_z"]), G["va_exact"], rtol=1e-12, atol=1e-15)
    assert np.allclose(_one_step(oracle, va, E, G["va_state"], G["va_z"]), G["va_euler"], rtol=1e-12, atol=1e-15)
    ci = CIRPPModel(0.0, "cp", cases.HAZARDS, 0.1, 0.01, 0.02, 1e-4)
    assert np.allclose(_one_step(oracle, ci, E, G["ci_state"], G["ci_z"]), G["ci_euler"], rtol=1e-12, atol=1e-16)
    cd = CIRPPModel(0.0, "cp", cases.HAZARDS, 0.1, 0.01, 0.02, 1e-4, deterministic=True)
    assert np.allclose(_one_step(oracle, cd, E, G["ci_state"], G["ci_z"]), G["cd_euler"], rtol=1e-12, atol=1e-16)
    assert np.allclose(np.array(cd._initial_state()), G["cd_init_state"][0])
    he = HestonModel(0.0, 800.0, 0.04, 0.45545583, -0.78975708, 0.01713417, 2.0, 0.0286834)
    assert np.allclose(_one_step(oracle, he, E, G["he_state"], G["he_z"]), G["he_euler"], rtol=1e-12, atol=1e-15)
    for smooth, tag in ((False, "hard"), (True, "fuzzy")):
        he.perform_smoothing = smooth
        got = _one_step(oracle, he, Q, G["he_state"], G["he_z"], G[f"he_qe_{tag}_u"][:, 0])
        assert np.allclose(got, G[f"he_qe_{tag}"], rtol=1e-11, atol=1e-14), tag
    he2 = HestonModel(0.0, 100.0, 0.02, 1.2, -0.5, 0.5, 0.02, 0.02)
    for smooth, tag in ((False, "hard"), (True, "fuzzy")):
        he2.perform_smoothing = smooth
        got = _one_step(oracle, he2, Q, G["he2_state"], G["he_z"], G[f"he2_qe_{tag}_u"][:, 0])
        assert np.allclose(got, G[f"he2_qe_{tag}"], rtol=1e-10, atol=1e-13), tag


def check_simulate_time_step_api(backend):
    """Model.simulate_time_step_{analytically,euler,qe}(time1, time2, state, corr_randn) — the reference's signature
    (models/vasicek.py:61-112, cirpp.py:174-198, heston.py:99-253) — through mcx_generate_paths_from_state on `backend`:
    per-path start states and caller-correlated normals, against the step maps recorded from the reference"""
    from mcx import _native
    A, E, Q = SimulationScheme.ANALYTICAL, SimulationScheme.EULER, SimulationScheme.QE
    t1, t2 = torch.tensor(float(G["t1"][0]), dtype=torch.float64), torch.tensor(float(G["t2"][0]), dtype=torch.float64)
    dt = float(t2 - t1)
    prev = _native._default_backend
    _native.set_backend(backend)
    try:
        def run(model, scheme, state, z, name, u=None):
            corr = torch.from_numpy(z) @ model.get_cholesky(scheme, dt).T
            st = torch.from_numpy(state)
            if scheme == Q:
                return model.simulate_time_step_qe(t1, t2, st, corr, torch.from_numpy(u)).cpu().numpy()
            fn = model.simulate_time_step_analytically if scheme == A else model.simulate_time_step_euler
            return fn(t1, t2, st, corr).cpu().numpy()
        bs = BlackScholesModel(0.0, 120.0, 0.05, 0.2)
        assert np.allclose(run(bs, A, G["bs_state"], G["bs_z"], "bs"), G["bs_exact"], rtol=1e-12)
        assert np.allclose(run(bs, E, G["bs_state"], G["bs_z"], "bs"), G["bs_euler"], rtol=1e-12)
        va = VasicekModel(0.0, 0.03, 0.05, 0.1, 0.01, asset_id="irs")
        assert np.allclose(run(va, A, G["va_state"], G["va_z"], "va"), G["va_exact"], rtol=1e-11, atol=1e-14)
        assert np.allclose(run(va, E, G["va_state"], G["va_z"], "va"), G["va_euler"], rtol=1e-11, atol=1e-14)
        ci = CIRPPModel(0.0, "cp", cases.HAZARDS, 0.1, 0.01, 0.02, 1e-4)
        assert np.allclose(run(ci, E, G["ci_state"], G["ci_z"], "ci"), G["ci_euler"], rtol=1e-11, atol=1e-15)
        he = HestonModel(0.0, 800.0, 0.04, 0.45545583, -0.78975708, 0.01713417, 2.0, 0.0286834)
        assert np.allclose(run(he, E, G["he_state"], G["he_z"], "he"), G["he_euler"], rtol=1e-11, atol=1e-14)
        got = run(he, Q, G["he_state"], G["he_z"], "he", G["he_qe_hard_u"])
        assert np.allclose(got, G["he_qe_hard"], rtol=1e-10, atol=1e-13)
    finally:
        _native.set_backend(prev)


def test_simulate_time_step_api_matches_reference(oracle):
    check_simulate_time_step_api(oracle)


def test_closed_forms_and_cholesky_match_reference():
    ci = CIRPPModel(0.0, "cp", cases.HAZARDS, 0.1, 0.01, 0.02, 1e-4)
    assert np.allclose([ci.psi(t) for t in G["ci_times"]], G["ci_psi"], rtol=1e-13)
    assert np.allclose([ci.cs_helper.probability_of_default(ci._hazards, ci._tenors, t) for t in G["ci_times"]], G["ci_pd"],
                       rtol=1e-13, atol=1e-16)
    y = torch.from_numpy(G["ci_state"][:, 0])
    for (a, b), ref in zip(G["ci_pairs"], G["ci_cond_survival"]):
        assert np.allclose(ci.survival_probability(a, b, y).numpy(), ref, rtol=1e-13)
    cd = CIRPPModel(0.0, "cp", cases.HAZARDS, 0.1, 0.01, 0.02, 1e-4, deterministic=True)
    for (a, b), ref in zip(G["ci_pairs"], G["cd_cond_survival"]):
        assert np.allclose(cd.survival_probability(a, b, y).numpy(), ref, rtol=1e-13)
    va = VasicekModel(0.0, 0.03, 0.05, 0.1, 0.01, asset_id="irs")
    r = torch.from_numpy(G["va_state"][:, 0])
    assert np.allclose(va.compute_bond_price(0.0, 2.0, r).numpy(), G["va_zcb_0_2"], rtol=1e-13)
    assert np.allclose(va.compute_bond_price(0.65, 3.0, r).numpy(), G["va_zcb_065_3"], rtol=1e-13)
    for rho in (-0.95, 0.0, 0.5, 0.99999):
        mc = cases.irs_models(rho)
        assert np.array_equal(mc.get_cholesky(SimulationScheme.EULER, None).numpy(), G[f"mc_chol_{rho}"])
    models = [BlackScholesModel(0.0, 100.0, 0.0, 0.4, asset_id=f"a{i}") for i in range(4)]
    mc = ModelConfig(models, inter_asset_correlation_matrix=np.array([[0.5]] * 6))
    assert np.allclose(mc.get_cholesky(SimulationScheme.EULER, None).numpy(), G["mc4_chol_euler"], rtol=1e-15)
    assert np.allclose(mc.get_cholesky(SimulationScheme.ANALYTICAL, 0.25).numpy(), G["mc4_chol_analytical"], rtol=1e-14)
    he = HestonModel(0.0, 800.0, 0.04, 0.45545583, -0.78975708, 0.01713417, 2.0, 0.0286834)
    assert np.allclose(he.get_cholesky(SimulationScheme.EULER, None).numpy(), G["he_chol_euler"], rtol=1e-15)


@pytest.mark.parametrize("tag,scheme,steps", [("exact", SimulationScheme.ANALYTICAL, 2), ("euler", SimulationScheme.EULER, 5)])
def test_model_config_correlated_paths(oracle, tag, scheme, steps):
    """engine-only: ModelConfig of 4 correlated Black-Scholes assets (model_config.py:101-276), recorded draws replayed"""
    g = np.load(os.path.join(cases.GOLDEN, "paths_mc4.npz"))
    models = [BlackScholesModel(0.0, 100.0 + 5 * i, 0.01 * i, 0.2 + 0.1 * i, asset_id=f"a{i}") for i in range(4)]
    mc = ModelConfig(models, inter_asset_correlation_matrix=np.array([[0.5], [0.3], [-0.2], [0.1], [0.4], [0.6]]))
    eng = MonteCarloEngine(g[f"{tag}_timeline"], scheme, mc, 512, steps, False, backend=oracle)
    eng.inject_z = torch.from_numpy(np.ascontiguousarray(np.transpose(g[f"{tag}_z"], (0, 2, 1))))
    p = eng.generate_paths().numpy()
    assert np.allclose(p, g[f"{tag}_paths"], rtol=1e-12)


def test_normal_equation_solver_matches_lstsq():
    rng = np.random.default_rng(3)
    x = 0.03 + 0.01 * rng.standard_normal(4096)
    Y = np.stack([0.5 * x - 0.01 + 0.02 * rng.standard_normal(4096), np.zeros(4096)], 1)
    K, S = 3, 2
    lo, hi = x.min(), x.max()
    shift, scale = 0.5 * (lo + hi), 2.0 / (hi - lo)
    z = (x - shift) * scale
    m = [np.sum(z ** k) for k in range(2 * K - 1)] + [np.sum(z ** k * Y[:, s]) for s in range(S) for k in range(K)]
    got = solve_normal_equations(np.array(m), K, S, shift, scale, False, lo)
    A = np.stack([x ** k for k in range(K)], 1)
    ref = torch.linalg.lstsq(torch.from_numpy(A), torch.from_numpy(Y)).solution.numpy().T
    assert np.allclose(got, ref, rtol=1e-7, atol=1e-10)
    assert np.allclose(A @ got.T, A @ ref.T, rtol=1e-10, atol=1e-13)
    # exactly rank-1 (all paths share x): minimum-norm solution, as gelsy returns at the calibration date
    x0 = 0.03
    A1 = np.tile([1.0, x0, x0 ** 2], (64, 1))
    y1 = rng.standard_normal(64)
    ref1 = torch.linalg.lstsq(torch.from_numpy(A1), torch.from_numpy(y1[:, None])).solution.numpy()[:, 0]
    m1 = [64.0] * (2 * K - 1) + [y1.sum()] * K
    got1 = solve_normal_equations(np.array(m1), K, 1, x0, 1.0, True, x0)[0]
    assert np.allclose(got1, ref1, rtol=1e-10)


def test_radix_select_driver_matches_sort(oracle):
    from mcx.parallel import Shard
    rng = np.random.default_rng(5)
    n, E = 5003, 6
    x = rng.standard_normal((E, n))
    x[1] = np.round(x[1], 1)
    x[2] = 0.0
    x[3, : n // 2] = -np.abs(x[3, : n // 2]) * 1e-300
    x[4] = -np.abs(x[4])
    for unsec in (UnsecuredSpec(np.arange(E), None, 0.0, False), UnsecuredSpec(np.arange(E), None, 0.3, False),
                  UnsecuredSpec(np.arange(E), np.array([-1, 0, 1, 2, 3, 4]), 0.1, True)):
        sc, _ = cases.make_controller("bs_european", oracle, inject=False)
        for q in (1, 2500, n - 2):
            vals = sc._select_order_stats(Shard(), unsec, torch.from_numpy(x), [q - 1, q, q + 1])
            ref = oracle.pfe_sort(unsec, torch.from_numpy(x), q)
            assert np.array_equal(vals, ref), q


def test_netting_algebra_literals():
    """the literal tensors of the reference's tests/pytests/test_netting_sets.py:209-264 through the oracle's prologue"""
    from oracle_backend import OracleBackend
    be = OracleBackend()
    expo = torch.tensor([[0.0, 0.0], [5.0, 10.0], [10.0, 20.0], [15.0, 30.0], [20.0, 40.0]], dtype=torch.float64)
    unsec = UnsecuredSpec(np.array([0, 2, 4]), np.array([-1, 1, 3]), 0.0, True)
    got = be.unsecured(unsec, expo).numpy()
    assert np.array_equal(got, np.array([[0.0, 0.0], [5.0, 10.0], [5.0, 10.0]]))
    thr = UnsecuredSpec(np.array([0, 1, 2]), None, 6.0, False)
    got = be.unsecured(thr, expo).numpy()
    assert np.array_equal(got, np.array([[0.0, 0.0], [0.0, 4.0], [4.0, 14.0]]))


def test_batched_normal_equation_solver_matches_scalar_one():
    """mcx.plan.solve_normal_equations_batch (product-batched LSM) against solve_normal_equations, incl. a degenerate system"""
    from mcx.plan import solve_normal_equations, solve_normal_equations_batch
    rng = np.random.default_rng(7)
    K, S, n_jobs, N = 3, 2, 6, 500
    moms, shifts, scales, degs, xmins = [], [], [], [], []
    for j in range(n_jobs):
        x = rng.normal(100.0, 10.0, N) if j != 3 else np.full(N, 42.0)
        lo, hi = x.min(), x.max()
        deg = not (hi > lo)
        shift, scale = (0.5 * (lo + hi), 2.0 / (hi - lo)) if not deg else (lo, 1.0)
        z = (x - shift) * scale
        y = rng.normal(size=(S, N)) + 0.01 * x
        m = [np.sum(z ** k) for k in range(2 * K - 1)] + [np.sum(z ** k * y[s]) for s in range(S) for k in range(K)]
        moms.append(m); shifts.append(shift); scales.append(scale); degs.append(deg); xmins.append(lo)
    moms = np.array(moms)
    batch = solve_normal_equations_batch(moms, K, S, np.array(shifts), np.array(scales), np.array(degs), np.array(xmins))
    for j in range(n_jobs):
        one = solve_normal_equations(moms[j], K, S, shifts[j], scales[j], degs[j], xmins[j])
        assert np.allclose(batch[j], one, rtol=1e-10, atol=1e-12), (j, batch[j], one)
