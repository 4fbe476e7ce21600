"""Black-Scholes single-asset GBM (reference: models/black_scholes.py:4-111).
params (gradient order) = [spot, volatility, rate]; state = [S]; one normal per sub-step."""
from __future__ import annotations

import math

import torch

from .. import _abi
from ..common.enums import SimulationScheme
from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequestType as RT
from .model import cexp, csqrt, AtomCoef, Model, SlotSpec


def deterministic_rate_atom(req, rate: float, t0: float) -> AtomCoef | None:
    """requests a flat deterministic curve answers (black_scholes.py:91-109, heston.py:261-278); note the reference's
    FORWARD_RATE is the growth factor exp(+r (t2-t1)) — reproduced as is."""
    k = req.request_type
    if k == RT.DISCOUNT_FACTOR:
        return AtomCoef(a=cexp(-rate * (req.time1 - t0)))
    if k == RT.FORWARD_RATE:
        return AtomCoef(a=cexp(rate * (req.time2 - req.time1)))
    if k == RT.LIBOR_RATE:
        return AtomCoef(a=(cexp(rate * (req.time2 - req.time1)) - 1) / (req.time2 - req.time1))
    if k == RT.NUMERAIRE:
        return AtomCoef(a=cexp(rate * (req.time1 - t0)))
    return None


class BlackScholesModel(Model):
    def __init__(self, calibration_date: float, spot: float, rate: float, sigma: float, asset_id: str | None = None):
        super().__init__(calibration_date=calibration_date, asset_ids=[asset_id] if asset_id else None)
        self.model_params = [torch.tensor(v, dtype=FLOAT, device=device) for v in (spot, sigma, rate)]

    def get_spot(self):
        return torch.stack([self.model_params[0]])

    def get_volatility(self):
        return torch.stack([self.model_params[1]])

    def get_rate(self):
        return torch.stack([self.model_params[2]])

    def get_model_param_names(self) -> list[str]:
        return ["spot", "volatility", "rate"]

    def _get_covariance_matrix(self, delta_t) -> torch.Tensor:
        sigma = self.get_volatility()
        return torch.diag(sigma * sigma * float(delta_t))            # black_scholes.py:44-48

    def _slots(self):
        return [SlotSpec(_abi.MODEL_BS, [self._pf(0), self._pf(1), self._pf(2)], 1, 1)]

    def _initial_state(self):
        return [self._pf(0)]

    def _step_aux(self, scheme, t1, dt):
        sigma, rate = self._pf(1), self._pf(2)
        if scheme == SimulationScheme.ANALYTICAL:
            return [[rate * dt, 0.5 * dt * sigma ** 2]]              # black_scholes.py:65-66
        return [[]]

    def _atom(self, req, asset_id):
        if req.request_type == RT.SPOT:
            return AtomCoef(col=0, d=1.0)
        at = deterministic_rate_atom(req, self._pf(2), self.t0())
        if at is None:
            raise NotImplementedError(f"Request type {req.request_type} not supported.")
        return at

    def _supports_scheme(self, scheme):
        return scheme in (SimulationScheme.ANALYTICAL, SimulationScheme.EULER)
