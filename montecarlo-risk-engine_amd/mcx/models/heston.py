"""Heston stochastic volatility (reference: models/heston.py:20-280).
params (gradient order) = [spot, volatility(vol-of-vol), rate, rho, kappa, theta, initial_variance];
state = [log S, v]; two normals (+ one uniform under QE) per sub-step."""
from __future__ import annotations

import math

import torch

from .. import _abi
from ..common.enums import SimulationScheme
from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequestType as RT
from .black_scholes import deterministic_rate_atom
from .model import AtomCoef, Model, SlotSpec


class HestonModel(Model):
    def __init__(self, calibration_date: float, spot: float, rate: float, sigma: float, rho: float, kappa: float,
                 theta: float, v0: float, asset_id: str | None = None):
        super().__init__(calibration_date=calibration_date, asset_ids=[asset_id] if asset_id else None,
                         simulation_dim=2, state_dim=2)
        self.model_params = [torch.tensor(v, dtype=FLOAT, device=device)
                             for v in (spot, sigma, rate, rho, kappa, theta, v0)]

    def get_spot(self):
        return torch.stack([self.model_params[0]])

    def get_volatility(self):
        return torch.stack([self.model_params[1]])

    def get_rate(self):
        return torch.stack([self.model_params[2]])

    def get_rho(self):
        return torch.stack([self.model_params[3]])

    def get_kappa(self):
        return torch.stack([self.model_params[4]])

    def get_theta(self):
        return torch.stack([self.model_params[5]])

    def get_initial_variance(self):
        return torch.stack([self.model_params[6]])

    def get_model_param_names(self) -> list[str]:
        return ["spot", "volatility", "rate", "rho", "kappa", "theta", "initial_variance"]

    def _get_correlation_matrix(self, simulation_scheme):
        if simulation_scheme == SimulationScheme.QE:                  # heston.py:85-90: QE draws independent normals
            return torch.eye(2, dtype=FLOAT, device=device)
        rho = self._pf(3)
        return torch.tensor([[1.0, rho], [rho, 1.0]], dtype=FLOAT, device=device)

    def _slots(self):
        flags = _abi.FLAG_SMOOTHING if self.perform_smoothing else 0
        return [SlotSpec(_abi.MODEL_HESTON, [self._pf(i) for i in range(7)], 2, 2, flags)]

    def _initial_state(self):
        return [math.log(self._pf(0)), self._pf(6)]

    def _n_uniform(self, scheme):
        return 1 if scheme == SimulationScheme.QE else 0

    def _step_aux(self, scheme, t1, dt):
        if scheme != SimulationScheme.QE:
            return [[]]
        sigma, rho, kappa, theta = self._pf(1), self._pf(3), self._pf(4), self._pf(5)
        E = math.exp(-kappa * dt)
        g1, g2 = 1.0, 0.0                                            # heston.py:151-152
        K0 = -rho * kappa * theta / sigma * dt
        K1 = (kappa * rho / sigma - 0.5) * g1 * dt - rho / sigma
        K2 = (kappa * rho / sigma - 0.5) * g2 * dt + rho / sigma
        K3 = (1.0 - rho * rho) * g1 * dt
        K4 = (1.0 - rho * rho) * g2 * dt
        c1 = sigma ** 2 * E * (1 - E) / kappa                        # heston.py:130
        c2 = theta * sigma ** 2 * (1 - E) ** 2 / (2 * kappa)         # heston.py:131
        return [[E, K0, K1, K2, K3, K4, c1, c2]]

    def _atom(self, req, asset_id):
        if req.request_type == RT.SPOT:
            return AtomCoef(col=0, b=1.0, c0=0.0, c1=1.0)            # exp(log S), heston.py:258-260
        at = deterministic_rate_atom(req, self._pf(2), self.t0())
        if at is None:
            raise NotImplementedError(f"Request type {req.request_type} not supported.")
        return at

    def _supports_scheme(self, scheme):
        return scheme in (SimulationScheme.QE, SimulationScheme.EULER)
