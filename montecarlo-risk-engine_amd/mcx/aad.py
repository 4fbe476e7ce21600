"""Sensitivities (`differentiate=True`) without a tape.

The reference puts every model parameter on torch's autograd tape and differentiates each metric value through the whole
simulation (controller/controller.py:609-648).  Here the derivative travels FORWARD with the path: the tangent kernel
(csrc/kt_tangent.hip) carries d state / d theta_j in registers and returns per-path d cashflow / d theta_j, so
d PV / d theta_j = mean_j — identical to the reference's pathwise gradient, including the smoothing it switches on
(`Model.requires_grad`, models/model.py:83-90) and torch's subgradient conventions.

Supported this round (BASELINE configs 2 and 4): PV metrics of European options on an Equity under a single
Black-Scholes or Heston model.  Sensitivities through the LSM regression (CVA/EPE greeks) are SURVEY §8f rank 1."""
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
