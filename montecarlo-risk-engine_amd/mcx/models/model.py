"""Base model: the host-side description of one SDE family.

Mirrors the reference's `Model` surface (models/model.py:5-141): calibration_date, asset_ids, model_params (list of
float64 scalar tensors in gradient order), simulation_dim / state_dim, perform_smoothing, get_cholesky with the same
caching rule (one factor per distinct dt for ANALYTICAL, one for EULER/QE).  The per-timestep evolution itself is NOT
done here: the reference's simulate_time_step_* chains run inside the HIP path kernel (csrc/k1_paths.hip); a model only
contributes its *slot description* (kind, parameters, per-sub-step constants) and the closed-form coefficients of the
requests it can resolve ("atoms", see include/mcx.h).
"""
from __future__ import annotations

import cmath
import math
from dataclasses import dataclass, field

import numpy as np
import torch

from ..common.enums import SimulationScheme
from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequest


def cexp(x):
    """exp for the closed-form coefficient code: real in, real out; complex in (complex-step differentiation of the
    descriptors, mcx/aad.py), complex out"""
    return cmath.exp(x) if isinstance(x, complex) else math.exp(x)


def csqrt(x):
    return cmath.sqrt(x) if isinstance(x, complex) else math.sqrt(x)


@dataclass
class SlotSpec:
    kind: int
    params: list
    state_dim: int
    sim_dim: int
    flags: int = 0


@dataclass
class AtomCoef:
    """value = a + d*x + b*exp(c0 + c1*x) with x = state[col] (col None: no state dependence)."""
    col: int | None = None
    a: float = 0.0
    d: float = 0.0
    b: float = 0.0
    c0: float = 0.0
    c1: float = 0.0

    def evaluate(self, state: torch.Tensor) -> torch.Tensor:
        """host evaluation on a [N, state_dim] (or [N]) tensor: API parity with Model.resolve_request"""
        if self.col is None:
            n = state.shape[0]
            return torch.full((n,), self.a, dtype=FLOAT, device=state.device)
        x = state[:, self.col] if state.ndim == 2 else state
        out = self.a + self.d * x
        if self.b != 0.0:
            out = out + self.b * torch.exp(self.c0 + self.c1 * x)
        return out


class Model:
    def __init__(self, calibration_date: float, simulation_dim: int = 1, state_dim: int = 1,
                 asset_ids: list[str] | None = None):
        if isinstance(calibration_date, torch.Tensor):
            calibration_date = float(calibration_date.reshape(-1)[0])
        self.calibration_date = torch.tensor([calibration_date], dtype=FLOAT, device=device)
        self.asset_ids: list[str] = asset_ids if asset_ids else [""]
        self.model_params: list[torch.Tensor] = []
        self.num_assets = len(self.asset_ids)
        self.simulation_dim = simulation_dim
        self.state_dim = state_dim
        self.perform_smoothing = False
        self._cholesky: dict = {}

    # ---- reference API -------------------------------------------------------------------------------------------
    def get_model_params(self):
        return self.model_params

    def get_model_param_names(self) -> list[str]:
        return [f"param_{i}" for i in range(len(self.model_params))]

    def requires_grad(self):
        """differentiate=True: smoothing on, parameters marked (reference: models/model.py:83-90). Sensitivities are
        produced by the tangent kernels, the flag only selects the smoothed primal exactly as the reference does."""
        self.perform_smoothing = True
        for p in self.model_params:
            p.requires_grad_(True)

    def _pf(self, i: int) -> float:
        """parameter i as a Python float; memoised on the tensor's identity and in-place version counter (the closed-form
        coefficient code calls this ~10^4 times per compilation)"""
        t = self.model_params[i]
        step = self.__dict__.get("_complex_step")
        if step is not None:                   # complex parameter values: the closed forms are evaluated at theta + i h
            return step[i]
        cache = self.__dict__.setdefault("_pf_cache", {})
        hit = cache.get(i)
        if hit is not None and hit[0] is t and hit[1] == t._version:
            return hit[2]
        v = float(t.detach())
        cache[i] = (t, t._version, v)
        return v

    def t0(self) -> float:
        return float(self.calibration_date[0])

    def _get_correlation_matrix(self, simulation_scheme: SimulationScheme) -> torch.Tensor:
        return torch.eye(self.simulation_dim, dtype=FLOAT, device=device)

    def _get_covariance_matrix(self, delta_t) -> torch.Tensor:
        return torch.eye(self.simulation_dim, dtype=FLOAT, device=device) * float(delta_t)

    def get_cholesky(self, simulation_scheme: SimulationScheme, delta_t) -> torch.Tensor:
        if simulation_scheme == SimulationScheme.ANALYTICAL:
            key = (simulation_scheme, float(delta_t))
            if key not in self._cholesky:
                self._cholesky[key] = torch.linalg.cholesky(self._get_covariance_matrix(float(delta_t)).detach())
            return self._cholesky[key]
        key = (simulation_scheme, None)
        if key not in self._cholesky:
            self._cholesky[key] = torch.linalg.cholesky(self._get_correlation_matrix(simulation_scheme).detach())
        return self._cholesky[key]

    def get_state(self, num_paths: int) -> torch.Tensor:
        s = torch.tensor(self._initial_state(), dtype=FLOAT, device=device)
        return s.unsqueeze(0).expand(num_paths, -1).clone()

    def resolve_request(self, req: AtomicRequest, asset_id: str, state: torch.Tensor) -> torch.Tensor:
        return self._atom(req, asset_id).evaluate(state)

    # ---- one time step from a caller-supplied state (models/*.py simulate_time_step_*) ---------------------------------
    def _simulate_time_step(self, scheme: SimulationScheme, time1, time2, state: torch.Tensor, corr_randn: torch.Tensor,
                            uniforms: torch.Tensor | None = None) -> torch.Tensor:
        """The reference's per-model step maps run inside the path kernel; this is that kernel on a one-step descriptor started
        from `state` [N, state_dim] (mcx_generate_paths_from_state) with the caller's ALREADY CORRELATED normals `corr_randn`
        [N, simulation_dim] injected (the descriptor's Cholesky factor is the identity)."""
        from .. import _native
        from ..plan import SimPlan
        be = _native.get_backend()
        t1, t2 = float(torch.as_tensor(time1).reshape(-1)[0]), float(torch.as_tensor(time2).reshape(-1)[0])
        plan = SimPlan.single_step(self, scheme, t1, t2)
        n = state.shape[0]
        st = be.from_numpy(np.ascontiguousarray(state.detach().cpu().numpy().T, dtype=np.float64))          # [D][N]
        z = be.from_numpy(np.ascontiguousarray(corr_randn.detach().cpu().numpy().T[None], dtype=np.float64))  # [1][n_z][N]
        u = None
        if plan.n_uniform:
            if uniforms is None:
                uniforms = torch.rand(n, 1, dtype=FLOAT)
            u = be.from_numpy(np.ascontiguousarray(uniforms.detach().cpu().numpy().reshape(1, n), dtype=np.float64))
        sim = be.sim_create(plan)
        out = be.generate_paths(sim, 0, 0, n, inject_z=z, inject_u=u, init_state=st)                         # [1][D][N]
        return out[0].T.to(state.device).contiguous()

    def simulate_time_step_analytically(self, time1, time2, state, corr_randn):
        return self._simulate_time_step(SimulationScheme.ANALYTICAL, time1, time2, state, corr_randn)

    def simulate_time_step_euler(self, time1, time2, state, corr_randn):
        return self._simulate_time_step(SimulationScheme.EULER, time1, time2, state, corr_randn)

    def simulate_time_step_qe(self, time1, time2, state, corr_randn, uniforms=None):
        return self._simulate_time_step(SimulationScheme.QE, time1, time2, state, corr_randn, uniforms)

    # ---- native hooks (overridden per family) -------------------------------------------------------------------
    def _slots(self) -> list[SlotSpec]:
        raise NotImplementedError

    def _initial_state(self) -> list[float]:
        raise NotImplementedError

    def _step_aux(self, scheme: SimulationScheme, t1: float, dt: float) -> list[list[float]]:
        """per slot: up to MCX_AUX host-precomputed constants for the sub-step [t1, t1+dt] (include/mcx.h)"""
        return [[] for _ in self._slots()]

    def _atom(self, req: AtomicRequest, asset_id: str) -> AtomCoef:
        raise NotImplementedError(f"Request type {req.request_type} not supported by {type(self).__name__}.")

    def _supports_scheme(self, scheme: SimulationScheme) -> bool:
        return True
