"""Sensitivities (`differentiate=True`) without a tape.

The reference puts every model parameter on torch's autograd tape and differentiates each metric value through the whole
simulation (controller/controller.py:609-648).  Here the derivative travels FORWARD with the path: the tangent kernel
(csrc/kt_tangent.hip) carries d state / d theta_j in registers and returns per-path d cashflow / d theta_j, so
d PV / d theta_j = mean_j — identical to the reference's pathwise gradient, including the smoothing it switches on
(`Model.requires_grad`, models/model.py:83-90) and torch's subgradient conventions.

Tangent kernels exist (BASELINE configs 2 and 4) for PV metrics of European options on an Equity under a single
Black-Scholes or Heston model.

Sensitivities through the LSM regression (CVA / PV of stateless books under BS / Vasicek / CIR++ Euler) run in forward mode too:
`run_with_tangent_book` drives csrc/kt_book.hip (dual paths -> dual normal equations -> dual book -> dual CVA).

Every other configuration (`run_with_bumps`): sensitivities through the LSM regression — CVA / EPE / PFE greeks, SURVEY §8f
rank 1, reference test `tests/pytests/test_cva_large_netting_set_aad_vs_fd.py` — are computed by CENTRAL DIFFERENCES WITH
COMMON RANDOM NUMBERS: the Philox counters (or the injected draws) are identical in the bumped runs, so the difference
quotient converges to the same pathwise derivative the reference's tape returns (including the dependence of the regression
coefficients on the parameters through the pre-simulation), with O(h^2) truncation and no sampling noise.  It costs 2P + 1
passes of the millisecond-scale hot path and serves what the forward-mode kernels do not cover (exercise products, PFE,
collateral, analytic exposures, non-Euler schemes).

Exercise products (American / Bermudan / FlexiCall): the reference's tape has NO gradient through the boolean
`should_exercise` (bermudan_option.py:122-128): its sensitivities hold the exercise policy fixed.  A plain bump would let paths
near the exercise boundary flip their decision and add jump / (2 h N) spikes, so the base run RECORDS every exercise decision
(pre-simulation roll and main simulation, mcx_book_set_exercise_replay) and the bumped pair REPLAYS them; pinned against the
reference's autograd gradients (tests/golden/bermudan_swaption_aad.npz, american_put_aad.npz)."""
from __future__ import annotations

import ctypes as C
import os
import math
import time

import numpy as np

from . import _abi
from .metrics.metric import MetricType, mean_and_error
from .models.black_scholes import BlackScholesModel
from .models.heston import HestonModel
from .plan import SimPlan


class TangentOption(C.Structure):
    _fields_ = [("t_idx", C.c_int32), ("netting_set", C.c_int32), ("strike", C.c_double), ("sign", C.c_double),
                ("numeraire", C.c_double), ("dnum_drate", C.c_double)]


def _check_supported(sc):
    from .products.equity import Equity
    from .products.european_option import EuropeanOption
    if not isinstance(sc.model, (BlackScholesModel, HestonModel)):
        raise NotImplementedError("differentiate=True: tangent kernels exist for a single BlackScholesModel or HestonModel")
    if any(m.metric_type != MetricType.PV or not m._native for m in sc.risk_metrics.metrics):
        raise NotImplementedError("differentiate=True: only PV metrics are differentiated in this round "
                                  "(exposure / CVA sensitivities go through the LSM regression: SURVEY §8f rank 1)")
    for p in sc.products:
        if not isinstance(p, EuropeanOption) or not isinstance(p.underlying, Equity):
            raise NotImplementedError("differentiate=True: European options on an Equity underlying only")
    if sc.requires_higher_order_derivatives:
        raise NotImplementedError("second-order derivatives are not implemented")
    if any(sc._can_skip_monte_carlo_for_product(p) for p in sc.products):
        raise NotImplementedError("analytically valued products are not Monte-Carlo PVs")
    if len(sc.netting_sets) > _abi.FUSED_MAX_NS:
        raise NotImplementedError(f"differentiate=True: at most {_abi.FUSED_MAX_NS} netting sets")


def run_with_tangents(sc):
    from .parallel import Shard
    _check_supported(sc)
    t0 = time.perf_counter()
    be = sc.backend
    shard = sc.shard_factory()
    model = sc.model
    plan = SimPlan(model, sc.simulation_timeline.numpy(), sc.simulation_scheme, sc.num_steps)
    sim = be.sim_create(plan)
    sc.sim_plan = plan
    tl = {float(t): i for i, t in enumerate(sc.simulation_timeline)}
    rate, t_cal = model._pf(2), model.t0()
    opts = (TangentOption * len(sc.products))()
    for k, p in enumerate(sc.products):
        T = float(p.exercise_date[0])
        num = math.exp(rate * (T - t_cal))
        opts[k].t_idx, opts[k].netting_set = tl[T], sc.product_to_netting_set_idx[k]
        opts[k].strike, opts[k].sign, opts[k].numeraire, opts[k].dnum_drate = p._K, p._sign(), num, (T - t_cal) * num
    off, n_local = shard.split(sc.num_paths_mainsim)
    inject = sc._inject.get("main", (None, None))
    P = len(model.get_model_params())
    cfs, dcfs = be.tangent_european(sim, opts, len(sc.netting_sets), P, 43, off, n_local, inject[0], inject[1])
    sc.last_state.update(cfs=cfs, dcfs=dcfs, paths=None, expo=None)
    results, grads = [], []
    for ns_i in range(len(sc.netting_sets)):
        pv = mean_and_error(shard.all_gather_np(be.reduce_vector(cfs[ns_i]).view(np.float64)))
        g = [mean_and_error(shard.all_gather_np(be.reduce_vector(dcfs[ns_i, j]).view(np.float64)))[0] for j in range(P)]
        results.append([[pv] for _ in sc.risk_metrics.metrics])
        grads.append([[tuple(g)] for _ in sc.risk_metrics.metrics])
    sc.timings = dict(total=time.perf_counter() - t0, tangent=True)
    return sc._package(results, grads, [])


# ---- central differences with common random numbers ---------------------------------------------------------------------
def _leaf_models(model):
    return list(model.models) if hasattr(model, "models") else [model]


def _set_param(model, j: int, value: float):
    """overwrite the j-th entry of `model.get_model_params()` (ModelConfig: concatenation in sub-model order)"""
    import torch
    model._cholesky = {}                       # factors of a covariance matrix depend on the parameters (ANALYTICAL scheme)
    for m in _leaf_models(model):
        m._cholesky = {}
    for m in _leaf_models(model):
        n = len(m.model_params)
        if j < n:
            m.model_params[j] = torch.tensor(float(value), dtype=m.model_params[j].dtype)
            if hasattr(model, "models"):       # ModelConfig keeps the concatenated list
                model.model_params = [p for sub in model.models for p in sub.get_model_params()]
            return
        j -= n
    raise IndexError("parameter index out of range")


def bump_size(theta: float) -> float:
    return 1e-5 * max(abs(theta), 1e-2)


def _clone_controller(sc, model, float32_cache, objects=None):
    """a controller on the same book under `model`.  objects: (netting_sets, risk_metrics) to build it on — a private deep copy by
    default; the bumped controllers of one differentiation run one after the other and share ONE copy (a run only reassigns the
    regression coefficients it produces; copying the product graph was 12 ms per controller)"""
    import copy
    from .controller.controller import SimulationController
    ns, rm = objects if objects is not None else (copy.deepcopy(sc.netting_sets), copy.deepcopy(sc.risk_metrics))
    clone = SimulationController(ns, model, rm, sc.num_paths_mainsim,
                                 sc.num_paths_presim, sc.num_steps, sc.simulation_scheme, differentiate=False,
                                 regression_function=sc.regression_function, backend=sc.backend, use_mfma=sc.use_mfma)
    for attr in ("seed_offset", "allow_fused", "main_plan", "materialize", "_inject"):
        setattr(clone, attr, getattr(sc, attr))
    clone.reference_float32_cf_cache = float32_cache
    return clone


def run_with_bumps(sc):
    import copy
    if sc.requires_higher_order_derivatives:
        raise NotImplementedError("second-order derivatives are not implemented")
    t0 = time.perf_counter()
    smoothing = getattr(sc.model, "perform_smoothing", False)      # the reference differentiates the smoothed payoffs
    base = _clone_controller(sc, sc.model, sc.reference_float32_cf_cache)
    # Exercise products: the reference's tape has no gradient through the boolean `should_exercise` (bermudan_option.py:122-128),
    # i.e. its sensitivities hold the exercise policy fixed.  The base run records every exercise decision (pre-simulation roll
    # and main simulation), the bumped runs replay them: no path flips its policy under the bump.
    replay = None
    if any(p.get_num_states() > 1 for p in sc.products) and hasattr(sc.backend, "book_set_exercise_replay"):
        replay = dict(mode=1, pre=None, main=None)
        base.exercise_replay = replay
    res0 = base.run_simulation()
    if replay is not None:
        base.backend.book_set_exercise_replay(base.book, 0, None)
    sc.sim_plan, sc.last_state = base.sim_plan, base.last_state
    theta = [float(p.detach()) for p in sc.model.get_model_params()]
    P = len(theta)
    vals = []                       # [param][sign] -> nested results
    shared = (copy.deepcopy(sc.netting_sets), copy.deepcopy(sc.risk_metrics))      # the bumped controllers' object graph
    for j in range(P):
        h = bump_size(theta[j])
        pair = []
        for sgn in (+1.0, -1.0):
            m = copy.deepcopy(sc.model)
            m.perform_smoothing = smoothing
            for leaf in _leaf_models(m):
                leaf.perform_smoothing = smoothing
            _set_param(m, j, theta[j] + sgn * h)
            # the float32 cashflow cache of the reference's LSM is a rounding artefact: differencing through it would only add
            # noise of size eps_f32 / h, so the bumped pair runs with the float64 cache
            bumped = _clone_controller(sc, m, False, shared)
            if replay is not None:
                bumped.exercise_replay = dict(mode=2, pre=replay["pre"], main=replay["main"])
            pair.append(bumped.run_simulation().results)
            if replay is not None:
                bumped.backend.book_set_exercise_replay(bumped.book, 0, None)
            bumped.release_device_buffers()
        vals.append((pair, h))
    grads = []
    for ns_i, per_metric in enumerate(res0.results):
        gm = []
        for m_i, evals in enumerate(per_metric):
            ge = []
            for e_i in range(len(evals)):
                ge.append(tuple((float(vals[j][0][0][ns_i][m_i][e_i][0]) - float(vals[j][0][1][ns_i][m_i][e_i][0])) / (2.0 * vals[j][1])
                                for j in range(P)))
            gm.append(ge)
        grads.append(gm)
    sc.timings = dict(total=time.perf_counter() - t0, tangent=False, bumped_passes=2 * P)
    return sc._package([[[tuple(v) for v in evals] for evals in per_metric] for per_metric in res0.results], grads, [])


def tangent_kernels_apply(sc) -> bool:
    try:
        _check_supported(sc)
        return True
    except NotImplementedError:
        return False


# ---- forward mode through the exposure path (csrc/kt_book.hip) ---------------------------------------------------------------
class _NoTangentForm(Exception):
    pass


def _host_descriptors(sc, model=None, dtype=np.float64):
    """every host-computed number the kernels consume (slot parameters, initial state, per-step tables, atom coefficients) of a
    COMPILED controller, re-evaluated under `model` (default: its own) WITHOUT touching the GPU.  Only the model-dependent
    numbers are recomputed: the per-step tables of the existing sub-step schedule and the closed-form coefficients of the
    existing atoms (requests, events and terms do not depend on the model parameters)."""
    model = sc.model if model is None else model
    sim, comp = sc.sim_plan, sc._comp
    specs = model._slots()
    slots = np.zeros((sim.n_slots, _abi.SLOT_NPARAM), dtype=dtype)
    for s, sp in enumerate(specs):
        slots[s, :len(sp.params)] = sp.params
    aux = np.zeros(sim.aux.shape, dtype=dtype)
    for k in range(sim.n_steps):
        for s, vals in enumerate(model._step_aux(sim.scheme, float(sim.steps["t1"][k]), float(sim.steps["dt"][k]))):
            aux[k, s, :len(vals)] = vals
    atoms = np.zeros((len(comp.atoms), 5), dtype=dtype)
    cols = []
    for q, src in enumerate(comp.atom_src):
        if src is None:
            atoms[q, 0] = comp.atoms[q][2]
            cols.append(-1)
            continue
        co = model._atom(*src)
        atoms[q] = (co.a, co.d, co.b, co.c0, co.c1)
        cols.append(-1 if co.col is None else co.col)
    shape = (tuple(cols), tuple(sp.kind for sp in specs))
    return dict(slots=slots, init=np.array(model._initial_state(), dtype=dtype), aux=aux, atoms=atoms), shape


def _set_complex_step(model, j: int, h: float):
    """evaluate the closed forms at theta_j + i h (Model._pf): j indexes `model.get_model_params()`"""
    for m in _leaf_models(model):
        n = len(m.model_params)
        m._complex_step = [complex(float(p.detach()), h if q == j else 0.0) for q, p in enumerate(m.model_params)]
        j -= n


def _solve_dual(mom: np.ndarray, K: int, shift: float, scale: float, degenerate: bool, n_par: int):
    """coefficients and their tangents in the RAW monomial basis from the dual moments of mcx_tangent_lsm:
    G c = r  =>  dc = G^-1 (dr - dG c)   (what autograd through torch.linalg.lstsq returns at full rank)"""
    m, dm = mom[0], mom[1:1 + n_par]
    n = m[0]
    c, dc = np.zeros(K), np.zeros((n_par, K))
    if n <= 0:
        return c, dc
    if degenerate:
        # all paths share x = x0 (t = calibration date): minimum-norm solution c = v ybar / (v.v), v = [1, x0, .., x0^(K-1)];
        # z = x - x0 is zero in value, its tangent is d x0
        x0 = shift
        ybar, dybar, dx0 = m[2 * K - 1] / n, dm[:, 2 * K - 1] / n, dm[:, 1] / n
        v = np.array([x0 ** k for k in range(K)])
        dv = np.array([k * x0 ** (k - 1) if k > 0 else 0.0 for k in range(K)])
        vv = float(v @ v)
        c = v * (ybar / vv)
        for q in range(n_par):
            dvq = dv * dx0[q]
            dc[q] = dvq * (ybar / vv) + v * (dybar[q] / vv) - v * (ybar * 2.0 * float(v @ dvq) / vv ** 2)
        return c, dc
    idx = np.arange(K)[:, None] + np.arange(K)[None, :]
    G, r = m[idx], m[2 * K - 1:2 * K - 1 + K]
    cz = np.linalg.solve(G, r)
    T = np.zeros((K, K))
    for k in range(K):
        for j in range(k + 1):
            T[j, k] = (scale ** k) * math.comb(k, j) * ((-shift) ** (k - j))
    c = T @ cz
    for q in range(n_par):
        dG, dr = dm[q][idx], dm[q][2 * K - 1:2 * K - 1 + K]
        dc[q] = T @ np.linalg.solve(G, dr - dG @ cz)
    return c, dc


def _solve_dual_states(mom: np.ndarray, K: int, S: int, shift: float, scale: float, degenerate: bool, n_par: int):
    """_solve_dual for the S right-hand sides of an exercise product (moments of mcx_tangent_lsm_step: sums of z^k, then of
    z^k Y_s per state s): c [S][K], dc [n_par][S][K]"""
    c, dc = np.zeros((S, K)), np.zeros((n_par, S, K))
    for s_ in range(S):
        sel = list(range(2 * K - 1)) + list(range(2 * K - 1 + s_ * K, 2 * K - 1 + (s_ + 1) * K))
        cs, dcs = _solve_dual(mom[:, sel], K, shift, scale, degenerate, n_par)
        c[s_], dc[:, s_, :] = cs, dcs
    return c, dc


def run_with_tangent_book(sc):
    """d PV / d theta and d CVA / d theta in forward mode through pre-simulation, regression and main simulation."""
    import copy
    from .parallel import Shard
    from .request_interface.request_types import AtomicRequest, AtomicRequestType
    if sc.requires_higher_order_derivatives:
        raise _NoTangentForm("second order")
    rm = sc.risk_metrics
    if sc.simulation_scheme.name != "EULER":
        raise _NoTangentForm("scheme")
    if any(m.metric_type not in (MetricType.PV, MetricType.CVA, MetricType.EPE, MetricType.ENE, MetricType.CE, MetricType.EEPE,
                                 MetricType.PFE) or not m._native for m in rm.metrics):
        raise _NoTangentForm("metric")
    if len(sc.products) > 64:
        raise _NoTangentForm("products")
    if any(p.get_num_states() > 1 for p in sc.products) and not hasattr(sc.backend, "tangent_lsm_step"):
        raise _NoTangentForm("exercise products")
    if any(sc._can_skip_monte_carlo_for_product(p) for p in sc.products):
        raise _NoTangentForm("analytic shortcuts")
    t0 = time.perf_counter()
    be, shard = sc.backend, sc.shard_factory()
    NP = _abi.TANGENT_NP
    base = _clone_controller(sc, sc.model, sc.reference_float32_cf_cache)
    res0 = base.run_simulation()
    if base.book_plan.n_basis > 4:
        raise _NoTangentForm("basis")
    theta = [float(p.detach()) for p in sc.model.get_model_params()]
    P = len(theta)
    smoothing = getattr(sc.model, "perform_smoothing", False)

    # derivative of every descriptor number (closed forms evaluated in float64 on the host): 4-point central differences with
    # a wide step — truncation O(h^4) ~ 1e-12, rounding eps/h ~ 1e-13 relative.  (CIR++ sensitivities are small residuals of
    # large cancelling terms: a 2-point formula with h = 1e-6 left 1e-5 relative noise on d CVA / d sigma_cir.)
    d0, shape0 = _host_descriptors(base)                 # `base` is compiled: its sub-step schedule and atoms are re-evaluated

    def bumped_model(j, value):
        m = copy.deepcopy(sc.model)
        m.perform_smoothing = smoothing
        _set_param(m, j, value)
        return m

    dd = {k: np.zeros(v.shape + (P,)) for k, v in d0.items()}
    # Complex step: the closed forms are analytic in the parameters, so d f / d theta = Im f(theta + i h) / h with h far below
    # rounding — ONE evaluation per parameter, no subtractive cancellation (exact to the last bits).  Models whose closed forms
    # are not complex-safe fall back to 4-point central differences with a wide step (truncation O(h^4) ~ 1e-12, rounding
    # eps/h ~ 1e-13 relative; a 2-point formula with h = 1e-6 left 1e-5 relative noise on d CVA / d sigma_cir).
    mode = os.environ.get("MCX_DESCRIPTOR_FD", "complex")
    if mode == "complex":
        try:
            for j in range(P):
                m = copy.deepcopy(sc.model)
                m.perform_smoothing = smoothing
                h = 1e-30 * max(abs(theta[j]), 1e-2)
                _set_complex_step(m, j, h)
                ev, shp = _host_descriptors(base, m, dtype=np.complex128)
                if shp != shape0:
                    raise _NoTangentForm("descriptor structure depends on the parameters")
                for k in dd:
                    dd[k][..., j] = ev[k].imag / h
        except TypeError:                     # a closed form written with real-only functions
            mode = "4"
    if mode != "complex":
        four_point = mode == "4"
        for j in range(P):
            h = (1e-3 if four_point else 1e-4) * max(abs(theta[j]), 1e-2)
            ev = {}
            for mult in ((1, -1, 2, -2) if four_point else (1, -1)):
                ev[mult], shp = _host_descriptors(base, bumped_model(j, theta[j] + mult * h))
                if shp != shape0:
                    raise _NoTangentForm("descriptor structure depends on the parameters")
            for k in dd:
                if four_point:
                    dd[k][..., j] = (8.0 * (ev[1][k] - ev[-1][k]) - (ev[2][k] - ev[-2][k])) / (12.0 * h)
                else:
                    dd[k][..., j] = (ev[1][k] - ev[-1][k]) / (2.0 * h)
    t_desc = time.perf_counter() - t0
    t_pre = t_main = 0.0
    book, sim, K = base.book, base._sim, base.book_plan.n_basis
    n_coeffs = len(base.book_plan.coeffs)
    n_ns, n_metrics = len(sc.netting_sets), len(rm.metrics)
    n_eval = [len(res0.results[0][m_i]) for m_i in range(n_metrics)]
    grads = [[[[0.0] * P for _ in range(n_eval[m_i])] for m_i in range(n_metrics)] for _ in range(n_ns)]     # [ns][metric][eval][param]
    jobs = [(p_i, p) for p_i, p in enumerate(sc.products) if p_i in base._mc_set and base._product_requires_regression(p)]
    has_exercise = any(p.get_num_states() > 1 for p in sc.products)
    base_coeffs = np.asarray(be.book_get_coeffs(base.book), dtype=np.float64)[:n_coeffs] if (has_exercise and n_coeffs) else None
    rows = base.metric_exposure_indices.numpy().astype(np.int32) if rm.requires_exposure_profiles() else np.zeros(0, dtype=np.int32)

    def mean_of(vec):
        return mean_and_error(shard.all_gather_np(be.reduce_vector(vec).view(np.float64)))[0]

    for c0 in range(0, P, NP):
        sel = list(range(c0, min(c0 + NP, P)))
        pad = lambda a: np.ascontiguousarray(np.concatenate([a[..., sel], np.zeros(a.shape[:-1] + (NP - len(sel),))], axis=-1))
        dslot, dinit, daux = pad(dd["slots"]), pad(dd["init"]), pad(dd["aux"])
        datoms = be.from_numpy(pad(dd["atoms"]))
        coeffs, dcoeffs = np.zeros(max(n_coeffs, 1)), np.zeros((max(n_coeffs, 1), NP))
        if has_exercise and n_coeffs:
            # exercise decisions are taken from the PRIMAL coefficients of the base run (the reference's tape has no gradient through
            # `should_exercise`, bermudan_option.py:122-128): every row starts from them, the exposure rows get their tangents below
            coeffs[:n_coeffs] = base_coeffs
        t1 = time.perf_counter()
        if jobs:
            off, n_pre = shard.split(sc.num_paths_presim)
            paths_pre, dpaths_pre = be.tangent_paths(sim, dslot, dinit, daux, 42 + sc.seed_offset, off, n_pre,
                                                     sc._inject.get("pre", (None, None))[0])
            plan = [(p_i, p, base._regression_schedule(p_i, p)) for p_i, p in jobs]
            plan = [(p_i, p, sched, base._regression_atoms(sched, p.asset_ids[0])) for p_i, p, sched in plan]
            x_ids = sorted({x for _, _, _, atoms in plan for _, x in atoms})
            g = shard.all_gather_np(be.lsm_stats(book, x_ids, paths_pre))
            lo, hi = g[:, :, 0].min(axis=0), g[:, :, 1].max(axis=0)
            x_range = {x: (lo[i], hi[i]) for i, x in enumerate(x_ids)}
            for p_i, p, sched, atoms in plan:
                S = p.get_num_states()
                if S > 1:
                    # backward induction with the cashflow cache of every hypothetical state rolled in dual numbers along the
                    # frozen policy (mcx_tangent_lsm_step); the coefficient tangents of the exposure rows feed the main pass
                    W, dW = be.zeros(S, n_pre), be.zeros(NP, S, n_pre)
                    for (t_reg, r0, r1, _prod_idx, expo_idx), (num, x) in zip(sched, atoms):
                        xmin, xmax = x_range[x]
                        degenerate = not (xmax > xmin)
                        shift = 0.5 * (xmin + xmax) if not degenerate else xmin
                        scale = 2.0 / (xmax - xmin) if not degenerate else 1.0
                        mom = shard.all_reduce_np(be.tangent_lsm_step(book, p_i, r0, r1, num, x, shift, scale, datoms, paths_pre,
                                                                      dpaths_pre, W, dW))
                        if expo_idx is None:
                            continue
                        c, dc = _solve_dual_states(mom, K, S, shift, scale, degenerate, NP)
                        o = base._expo_coeff_base[p_i] + expo_idx * S * K
                        dcoeffs[o:o + S * K] = dc.reshape(NP, S * K).T
                    del W, dW
                    continue
                pdates = np.asarray([float(t) for t in p.product_timeline])
                for (t_reg, _r0, _r1, _prod_idx, expo_idx), (num, x) in zip(sched, atoms):
                    if expo_idx is None:
                        continue
                    xmin, xmax = x_range[x]
                    degenerate = not (xmax > xmin)
                    shift = 0.5 * (xmin + xmax) if not degenerate else xmin
                    scale = 2.0 / (xmax - xmin) if not degenerate else 1.0
                    first = int(np.searchsorted(pdates, t_reg, side="right"))      # cashflows strictly after t_reg (controller.py:323)
                    mom = shard.all_reduce_np(be.tangent_lsm(book, p_i, first, num, x, shift, scale, datoms, paths_pre, dpaths_pre))
                    c, dc = _solve_dual(mom, K, shift, scale, degenerate, NP)
                    o = base._expo_coeff_base[p_i] + expo_idx * K
                    coeffs[o:o + K], dcoeffs[o:o + K] = c, dc.T
            del paths_pre, dpaths_pre
        be.synchronize()
        t2 = time.perf_counter()
        t_pre += t2 - t1
        off, n_main = shard.split(sc.num_paths_mainsim)
        paths, dpaths = be.tangent_paths(sim, dslot, dinit, daux, 43 + sc.seed_offset, off, n_main,
                                         sc._inject.get("main", (None, None))[0])
        # analytic Black-Scholes exposure events: which tangent slot of this pass is their sigma / their rate
        ev_param = None
        ev_kind = base.book_plan.events["kind"]
        if (ev_kind == _abi.EV_EXPO_BS).any():
            ev_param = np.full((len(ev_kind), 2), -1, dtype=np.int32)
            for p_i, p in enumerate(sc.products):
                if hasattr(p, "_bs_param_indices") and base._can_use_analytic_exposure_for_product(p):
                    _i_s, i_v, i_r = p._bs_param_indices(sc.model)
                    b0, b1 = int(base.book_plan.products["ev_begin"][p_i]), int(base.book_plan.products["ev_end"][p_i])
                    rows_bs = np.arange(b0, b1)[ev_kind[b0:b1] == _abi.EV_EXPO_BS]
                    ev_param[rows_bs, 0] = sel.index(i_v) if i_v in sel else -1
                    ev_param[rows_bs, 1] = sel.index(i_r) if i_r in sel else -1
        cfs, expo = be.tangent_eval(book, datoms, be.from_numpy(coeffs), be.from_numpy(dcoeffs), paths, dpaths, ev_param)
        for ns_i, ns in enumerate(sc.netting_sets):
            prof = None
            coll = ns.is_collateralized()
            delayed = base.netting_set_delayed_exposure_indices[ns_i].numpy().astype(np.int32) if coll else None
            for m_i, m in enumerate(rm.metrics):
                if m.metric_type == MetricType.PV:
                    for q, j in enumerate(sel):
                        grads[ns_i][m_i][0][j] = mean_of(cfs[1 + q, ns_i])
                elif m.metric_type in (MetricType.EPE, MetricType.ENE, MetricType.CE, MetricType.EEPE):
                    if prof is None:            # [dates][2][NP] sums of 1[u>0] du / 1[u<0] du over all ranks
                        prof = shard.all_reduce_np(be.tangent_profiles(rows, ns.threshold, expo, ns_i, delayed, coll)) / float(sc.num_paths_mainsim)
                    for q, j in enumerate(sel):
                        if m.metric_type == MetricType.CE:           # positive part of the first exposure date (ce_metric.py)
                            grads[ns_i][m_i][0][j] = float(prof[0, 0, q])
                        elif m.metric_type == MetricType.EEPE:       # plain time average of the EE profile (eepe_metric.py)
                            grads[ns_i][m_i][0][j] = float(prof[:, 0, q].mean())
                        else:
                            side = 0 if m.metric_type == MetricType.EPE else 1
                            for e_i in range(n_eval[m_i]):
                                grads[ns_i][m_i][e_i][j] = float(prof[e_i, side, q])
                elif m.metric_type == MetricType.PFE:
                    # the tangent of the path that realises the exact order statistic (the reference differentiates through
                    # torch.sort): radix select on the primal image, then the first path carrying that value
                    from .plan import UnsecuredSpec
                    unsec = UnsecuredSpec(rows, delayed, ns.threshold, coll)
                    target = base._select_order_stats(shard, unsec, expo[0, ns_i], [m.q_index(sc.num_paths_mainsim)])[:, 0]
                    pick = be.tangent_pick(rows, ns.threshold, target, expo, ns_i, delayed, coll)
                    pick[:, 0] = np.where(pick[:, 0] >= 0, pick[:, 0] + off, np.inf)          # global path index
                    allp = shard.all_gather_np(pick)                                          # [world][dates][1+NP]
                    win = np.argmin(allp[:, :, 0], axis=0)
                    for e_i in range(n_eval[m_i]):
                        for q, j in enumerate(sel):
                            grads[ns_i][m_i][e_i][j] = float(allp[win[e_i], e_i, 1 + q])
                elif not (ns.counterparty_id is not None and m.counterparty_id != ns.counterparty_id):
                    surv, cond = base._cva_atoms[m_i]
                    out = be.tangent_cva(book, datoms, rows, surv, cond, ns.threshold, m.recovery_rate, expo, ns_i, paths, dpaths, delayed, coll)
                    for q, j in enumerate(sel):
                        grads[ns_i][m_i][0][j] = mean_of(out[1 + q])
        del paths, dpaths, cfs, expo
        be.synchronize()
        t_main += time.perf_counter() - t2
    sc.sim_plan, sc.last_state = base.sim_plan, base.last_state
    sc.timings = dict(total=time.perf_counter() - t0, tangent=True, forward_mode_passes=(P + NP - 1) // NP,
                      base_run_and_descriptor_derivatives=t_desc, presim_and_regression=t_pre, main=t_main)
    g = [[[tuple(ev) for ev in grads[ns_i][m_i]] for m_i in range(n_metrics)] for ns_i in range(n_ns)]
    return sc._package([[[tuple(v) for v in evals] for evals in per_metric] for per_metric in res0.results], g, [])


# ---- analytically evaluated PV metrics: autograd on the closed form (host scalars, no simulation) --------------------------------
def analytic_controller(sc) -> bool:
    return bool(sc.products) and all(sc._can_skip_monte_carlo_for_product(p) and hasattr(p, "compute_pv_analytically_torch")
                                     for p in sc.products)


def run_analytic_with_autograd(sc):
    """every product is valued by its closed form (PVMetric(ANALYTICAL)): first and, on request, second derivatives come from
    torch.autograd on that closed form, as in the reference (controller.py:609-648; tests/pytests/test_european_option_hessian.py)"""
    import torch
    t0 = time.perf_counter()
    theta = [torch.tensor(float(p.detach()), dtype=torch.float64, requires_grad=True) for p in sc.model.get_model_params()]
    P = len(theta)
    results, grads, hess = [], [], []
    for ns_i, ns in enumerate(sc.netting_sets):
        value = sum(p.compute_pv_analytically_torch(sc.model, theta) for p in ns.products)
        g = torch.autograd.grad(value, theta, create_graph=sc.requires_higher_order_derivatives, allow_unused=True)
        g = [torch.zeros((), dtype=torch.float64) if x is None else x for x in g]
        row = tuple(float(x.detach()) for x in g)
        H = None
        if sc.requires_higher_order_derivatives:
            H = []
            for x in g:
                if x.requires_grad:
                    hx = torch.autograd.grad(x, theta, retain_graph=True, allow_unused=True)
                    H.append(tuple(0.0 if y is None else float(y) for y in hx))
                else:
                    H.append(tuple(0.0 for _ in range(P)))
        results.append([[(float(value.detach()), 0.0)] for _ in sc.risk_metrics.metrics])
        grads.append([[row] for _ in sc.risk_metrics.metrics])
        hess.append([[tuple(H)] if H is not None else [] for _ in sc.risk_metrics.metrics])
    sc.timings = dict(total=time.perf_counter() - t0, analytic=True)
    return sc._package(results, grads, hess if sc.requires_higher_order_derivatives else [])


# ---- second-order derivatives of Monte-Carlo metrics (controller.py:253-255, 631-648) --------------------------------------------
def run_second_order(sc):
    """`compute_higher_derivatives()` with Monte-Carlo metrics: the Hessian as central differences, with common random numbers, of
    the first-order sensitivities this module produces (forward-mode kernels where they exist, replayed bumps otherwise):
    H[i][j] = (g_i(theta + h e_j) - g_i(theta - h e_j)) / 2h, 2P first-order runs.

    What it estimates: the derivative of the pathwise gradient INCLUDING the paths that cross a payoff kink under the bump — the
    term that makes a Monte-Carlo gamma non-zero.  The reference differentiates its tape a second time (`torch.autograd.grad` of
    the gradient, controller.py:631-648), which sees d/dx 1[x > 0] = 0 and therefore returns only the smooth part (a
    Black-Scholes call has gamma exactly 0 on its tape); for parameters that enter smoothly the two agree.  The reference's own
    tests request second derivatives of analytically valued PVs only (tests/pytests/test_european_option_hessian.py), served by
    `run_analytic_with_autograd`.  Bump: 1 % of the parameter (the kink term's variance grows like 1 / (N h))."""
    import copy
    from .controller.controller import SimulationController
    t0 = time.perf_counter()
    theta = [float(p.detach()) for p in sc.model.get_model_params()]
    P = len(theta)

    def first_order(model):
        c = SimulationController(copy.deepcopy(sc.netting_sets), model, copy.deepcopy(sc.risk_metrics), sc.num_paths_mainsim,
                                 sc.num_paths_presim, sc.num_steps, sc.simulation_scheme, differentiate=True,
                                 regression_function=sc.regression_function, backend=sc.backend, use_mfma=sc.use_mfma)
        for attr in ("seed_offset", "allow_fused", "main_plan", "materialize", "_inject", "forward_mode"):
            setattr(c, attr, getattr(sc, attr))
        return c, c.run_simulation()

    base, res0 = first_order(sc.model)
    sc.sim_plan, sc.last_state = base.sim_plan, base.last_state
    cols = []                                       # cols[j][ns][metric][eval] = tuple over i of d g_i / d theta_j
    for j in range(P):
        h = 1e-2 * max(abs(theta[j]), 1e-2)
        g = []
        for sgn in (+1.0, -1.0):
            m = copy.deepcopy(sc.model)
            m.perform_smoothing = getattr(sc.model, "perform_smoothing", False)
            for leaf in _leaf_models(m):
                leaf.perform_smoothing = m.perform_smoothing
            _set_param(m, j, theta[j] + sgn * h)
            g.append(first_order(m)[1].derivatives)
        cols.append([[[tuple((float(a) - float(b)) / (2.0 * h) for a, b in zip(ep, em)) for ep, em in zip(mp, mm)]
                      for mp, mm in zip(np_, nm_)] for np_, nm_ in zip(g[0], g[1])])
    higher = [[[tuple(tuple(cols[j][ns_i][m_i][e_i][i] for j in range(P)) for i in range(P))
                for e_i in range(len(res0.derivatives[ns_i][m_i]))]
               for m_i in range(len(res0.derivatives[ns_i]))] for ns_i in range(len(res0.derivatives))]
    sc.timings = dict(total=time.perf_counter() - t0, second_order_by_differences=True, first_order_runs=2 * P + 1)
    return sc._package([[[tuple(v) for v in evals] for evals in per_metric] for per_metric in res0.results],
                       [[[tuple(ev) for ev in per_metric] for per_metric in per_ns] for per_ns in res0.derivatives], higher)
