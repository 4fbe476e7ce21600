"""Sensitivities (`differentiate=True`) without a tape.

The reference puts every model parameter on torch's autograd tape and differentiates each metric value through the whole
simulation (controller/controller.py:609-648).  Here the derivative travels FORWARD with the path: the tangent kernel
(csrc/kt_tangent.hip) carries d state / d theta_j in registers and returns per-path d cashflow / d theta_j, so
d PV / d theta_j = mean_j — identical to the reference's pathwise gradient, including the smoothing it switches on
(`Model.requires_grad`, models/model.py:83-90) and torch's subgradient conventions.

Tangent kernels exist (BASELINE configs 2 and 4) for PV metrics of European options on an Equity under a single
Black-Scholes or Heston model.

Every other configuration (`run_with_bumps`): sensitivities through the LSM regression — CVA / EPE / PFE greeks, SURVEY §8f
rank 1, reference test `tests/pytests/test_cva_large_netting_set_aad_vs_fd.py` — are computed by CENTRAL DIFFERENCES WITH
COMMON RANDOM NUMBERS: the Philox counters (or the injected draws) are identical in the bumped runs, so the difference
quotient converges to the same pathwise derivative the reference's tape returns (including the dependence of the regression
coefficients on the parameters through the pre-simulation), with O(h^2) truncation and no sampling noise.  It costs 2P + 1
passes of the millisecond-scale hot path instead of one tangent pass; a forward-mode kernel for this path is the next step."""
from __future__ import annotations

import ctypes as C
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
    if len(sc.netting_sets) > _abi.FUSED_MAX_NS:
        raise NotImplementedError(f"differentiate=True: at most {_abi.FUSED_MAX_NS} netting sets")


def run_with_tangents(sc):
    from .parallel import Shard
    _check_supported(sc)
    t0 = time.perf_counter()
    be = sc.backend
    shard = Shard()
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
    for m in _leaf_models(model):
        n = len(m.model_params)
        if j < n:
            m.model_params[j] = torch.tensor(float(value), dtype=m.model_params[j].dtype)
            return
        j -= n
    raise IndexError("parameter index out of range")


def bump_size(theta: float) -> float:
    return 1e-5 * max(abs(theta), 1e-2)


def _clone_controller(sc, model, float32_cache):
    import copy
    from .controller.controller import SimulationController
    clone = SimulationController(copy.deepcopy(sc.netting_sets), model, copy.deepcopy(sc.risk_metrics), sc.num_paths_mainsim,
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
    res0 = base.run_simulation()
    sc.sim_plan, sc.last_state = base.sim_plan, base.last_state
    theta = [float(p.detach()) for p in sc.model.get_model_params()]
    P = len(theta)
    vals = []                       # [param][sign] -> nested results
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
            pair.append(_clone_controller(sc, m, False).run_simulation().results)
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
