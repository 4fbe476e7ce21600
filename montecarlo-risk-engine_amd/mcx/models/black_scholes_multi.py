"""n-asset Black-Scholes with its own correlation matrix (reference: models/black_scholes_multi.py:6-128).
params (gradient order) = [spots..., volatilities..., rate]; state = the n spots; n correlated normals per sub-step.

On the GPU this is n Black-Scholes slots of the path kernel sharing one Cholesky factor (csrc/k1_paths.hip): the factor of
the correlation matrix under EULER, of the covariance S rho S dt under ANALYTICAL — the reference's
`generate_correlated_randn` (model.py:38-73) with `_get_correlation_matrix` / `_get_covariance_matrix` (:52-61)."""
from __future__ import annotations

import numpy as np
import torch

from .. import _abi
from ..common.enums import SimulationScheme
from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequestType as RT
from .black_scholes import deterministic_rate_atom
from .model import AtomCoef, Model, SlotSpec


class BlackScholesMulti(Model):
    def __init__(self, calibration_date: float, rate: float, asset_ids: list[str], spots, volatilities, correlation_matrix):
        assert len(asset_ids) == len(spots) == len(volatilities)
        super().__init__(calibration_date=calibration_date, simulation_dim=len(asset_ids), state_dim=len(spots),
                         asset_ids=list(asset_ids))
        self.model_params = [torch.tensor(float(v), dtype=FLOAT, device=device)
                             for v in list(spots) + list(volatilities) + [rate]]
        self.correlation_matrix = torch.tensor(np.asarray(correlation_matrix), dtype=FLOAT, device=device)
        assert self.correlation_matrix.shape == (self.num_assets, self.num_assets)

    def get_spot(self):
        return torch.stack(self.model_params[:self.num_assets])

    def get_volatility(self):
        return torch.stack(self.model_params[self.num_assets:2 * self.num_assets])

    def get_rate(self):
        return self.model_params[2 * self.num_assets]

    def get_model_param_names(self) -> list[str]:
        return [*[f"spot[{a}]" for a in self.asset_ids], *[f"volatility[{a}]" for a in self.asset_ids], "rate"]

    def _get_correlation_matrix(self, simulation_scheme) -> torch.Tensor:
        return self.correlation_matrix

    def _get_covariance_matrix(self, delta_t) -> torch.Tensor:
        # S C S is the same for every step length: kept per volatility vector (an analytical-scheme timeline of a large book asks for
        # the factor of ~400 distinct step lengths; torch.diag + two products per call were 0.1 s of its planning)
        vols = tuple(self._pf(self.num_assets + i) for i in range(self.num_assets))
        hit = self.__dict__.get("_scs")
        if hit is None or hit[0] != vols:
            S = torch.diag(self.get_volatility().detach())
            hit = self.__dict__["_scs"] = (vols, S @ self.correlation_matrix @ S)
        return hit[1] * float(delta_t)                                       # black_scholes_multi.py:56-61

    # ---- native hooks -------------------------------------------------------------------------------------------
    def _rate(self) -> float:
        return self._pf(2 * self.num_assets)

    def _slots(self):
        n = self.num_assets
        return [SlotSpec(_abi.MODEL_BS, [self._pf(i), self._pf(n + i), self._rate()], 1, 1) for i in range(n)]

    def _initial_state(self):
        return [self._pf(i) for i in range(self.num_assets)]

    def _step_aux(self, scheme, t1, dt):
        n = self.num_assets
        if scheme == SimulationScheme.ANALYTICAL:                               # black_scholes_multi.py:75-79
            return [[self._rate() * dt, 0.5 * dt * self._pf(n + i) ** 2] for i in range(n)]
        return [[] for _ in range(n)]

    def _atom(self, req, asset_id):
        if req.request_type == RT.SPOT:
            return AtomCoef(col=self.asset_ids.index(asset_id), d=1.0)
        at = deterministic_rate_atom(req, self._rate(), self.t0())
        if at is None:
            raise NotImplementedError(f"Request type {req.request_type} not supported.")
        return at

    def _supports_scheme(self, scheme):
        return scheme in (SimulationScheme.ANALYTICAL, SimulationScheme.EULER)
