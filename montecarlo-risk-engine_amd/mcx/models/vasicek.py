"""Vasicek one-factor short rate (reference: models/vasicek.py:5-156).
params (gradient order) = [rate, volatility, mean, mean_reversion_speed]; state = [r, log B] (B = money-market account)."""
from __future__ import annotations

import math

import torch

from .. import _abi
from ..common.enums import SimulationScheme
from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequestType as RT
from .model import cexp, csqrt, AtomCoef, Model, SlotSpec


class VasicekModel(Model):
    def __init__(self, calibration_date: float, rate: float, mean: float, mean_reversion_speed: float,
                 volatility: float, asset_id: str | None = None):
        super().__init__(calibration_date=calibration_date, state_dim=2, asset_ids=[asset_id])
        self.model_params = [torch.tensor(v, dtype=FLOAT, device=device)
                             for v in (rate, volatility, mean, mean_reversion_speed)]

    def get_rate(self):
        return torch.stack([self.model_params[0]])

    def get_volatility(self):
        return torch.stack([self.model_params[1]])

    def get_mean(self):
        return torch.stack([self.model_params[2]])

    def get_mean_reversion_speed(self):
        return torch.stack([self.model_params[3]])

    def get_model_param_names(self) -> list[str]:
        return ["rate", "volatility", "mean", "mean_reversion_speed"]

    def _get_covariance_matrix(self, delta_t) -> torch.Tensor:
        sigma, a = self.get_volatility(), self.get_mean_reversion_speed()
        decay = torch.exp(-a * float(delta_t))
        return torch.diag((sigma ** 2 / (2 * a)) * (1 - decay ** 2))   # vasicek.py:52-59

    def _zcb_coeffs(self, time1: float, time2: float) -> tuple[float, float]:
        """P(t1,t2 | r) = exp(alpha - B r)  (vasicek.py:114-128)"""
        sigma, theta, a = self._pf(1), self._pf(2), self._pf(3)
        tau = time2 - time1
        B = (1 - cexp(-a * tau)) / a
        alpha = (theta - sigma ** 2 / (2 * a ** 2)) * (B - tau) - (sigma ** 2 / (4 * a)) * B ** 2
        return alpha, B

    def compute_bond_price(self, time1, time2, rate):
        alpha, B = self._zcb_coeffs(float(time1), float(time2))
        rate = torch.as_tensor(rate, dtype=FLOAT, device=device)
        return cexp(alpha) * torch.exp(-B * rate)

    def _slots(self):
        return [SlotSpec(_abi.MODEL_VASICEK, [self._pf(i) for i in range(4)], 2, 1)]

    def _initial_state(self):
        return [self._pf(0), 0.0]

    def _step_aux(self, scheme, t1, dt):
        if scheme == SimulationScheme.ANALYTICAL:
            return [[cexp(-self._pf(3) * dt)]]                     # vasicek.py:82
        return [[]]

    def _atom(self, req, asset_id):
        k = req.request_type
        if k == RT.SPOT:
            return AtomCoef(col=0, d=1.0)
        if k == RT.DISCOUNT_FACTOR:
            alpha, B = self._zcb_coeffs(self.t0(), req.time1)
            return AtomCoef(col=0, b=1.0, c0=alpha, c1=-B)
        if k == RT.FORWARD_RATE:
            alpha, B = self._zcb_coeffs(req.time1, req.time2)
            return AtomCoef(col=0, b=1.0, c0=alpha, c1=-B)
        if k == RT.LIBOR_RATE:                                         # (1/P - 1)/tau, vasicek.py:149-153
            alpha, B = self._zcb_coeffs(req.time1, req.time2)
            tau = req.time2 - req.time1
            return AtomCoef(col=0, a=-1.0 / tau, b=1.0 / tau, c0=-alpha, c1=B)
        if k == RT.NUMERAIRE:
            return AtomCoef(col=1, b=1.0, c0=0.0, c1=1.0)              # exp(log B)
        raise NotImplementedError(f"Request type {k} not supported by Vasicek.")

    def _supports_scheme(self, scheme):
        return scheme in (SimulationScheme.ANALYTICAL, SimulationScheme.EULER)
