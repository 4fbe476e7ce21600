"""CIR++ default-intensity model lambda(t) = y(t) + psi(t) (reference: models/cirpp.py:6-317).
params (gradient order) = [kappa, theta, sigma, y0]; state = [y, Lambda = int_0^t lambda]."""
from __future__ import annotations

import math

import torch

from .. import _abi
from ..common.enums import SimulationScheme
from ..common.packages import FLOAT, device
from ..helpers.cs_helper import CSHelper
from ..request_interface.request_types import AtomicRequestType as RT
from .model import cexp, csqrt, AtomCoef, Model, SlotSpec


class CIRPPModel(Model):
    def __init__(self, calibration_date: float, asset_id: str, hazard_rates: dict[float, float], kappa: float,
                 theta: float, volatility: float, y0: float, deterministic: bool = False):
        super().__init__(calibration_date=calibration_date, state_dim=2, asset_ids=[asset_id])
        assert 2 * kappa * theta - volatility ** 2 > 0 and y0 > 0, "Feller condition not met."
        self.model_params = [torch.tensor(v, dtype=FLOAT, device=device) for v in (kappa, theta, volatility, y0)]
        self.tenors = torch.tensor(list(hazard_rates.keys()), dtype=FLOAT, device=device)
        self.hazard_rates = torch.tensor(list(hazard_rates.values()), dtype=FLOAT, device=device)
        self._tenors = [float(t) for t in hazard_rates.keys()]
        self._hazards = [float(h) for h in hazard_rates.values()]
        self.deterministic = deterministic
        self.cs_helper = CSHelper()

    def get_kappa(self):
        return torch.stack([self.model_params[0]])

    def get_theta(self):
        return torch.stack([self.model_params[1]])

    def get_sigma(self):
        return torch.stack([self.model_params[2]])

    def get_y0(self):
        return torch.stack([self.model_params[3]])

    def get_model_param_names(self) -> list[str]:
        return ["kappa", "theta", "sigma", "y0"]

    # ---- market curve (cirpp.py:72-89) --------------------------------------------------------------------------
    def _lambda_market(self, t) -> float:
        t = float(t)
        for idx, tenor in enumerate(self._tenors):
            if t <= tenor:
                return self._hazards[idx]
        return self._hazards[-1]

    def _market_survival_probability(self, t) -> float:
        return 1.0 - self.cs_helper.probability_of_default(self._hazards, self._tenors, float(t))

    # ---- CIR closed forms (cirpp.py:93-142) ---------------------------------------------------------------------
    def _h(self) -> float:
        kappa, sigma = self._pf(0), self._pf(2)
        return csqrt(kappa * kappa + 2.0 * sigma * sigma)

    def _A(self, t, T) -> float:
        kappa, theta, sigma, h = self._pf(0), self._pf(1), self._pf(2), self._h()
        dt = float(T) - float(t)
        num = 2.0 * h * cexp(0.5 * (kappa + h) * dt)
        den = 2.0 * h + (kappa + h) * (cexp(h * dt) - 1.0)
        return (num / den) ** ((2.0 * kappa * theta) / (sigma * sigma))

    def _B(self, t, T) -> float:
        kappa, h = self._pf(0), self._h()
        dt = float(T) - float(t)
        e = cexp(h * dt) - 1.0
        return (2.0 * e) / (2.0 * h + (kappa + h) * e)

    def _D(self, t) -> float:
        kappa, theta, sigma, h = self._pf(0), self._pf(1), self._pf(2), self._h()
        et = cexp(h * float(t))
        num = 0.5 * (kappa + h) - (h * (kappa + h) * et) / (2.0 * h + (kappa + h) * (et - 1.0))
        return (2.0 * kappa * theta / (sigma * sigma)) * num

    def _E(self, t) -> float:
        kappa, h = self._pf(0), self._h()
        et = cexp(h * float(t))
        return (4.0 * h * h * et) / (2.0 * h + (kappa + h) * (et - 1.0)) ** 2

    def psi(self, t) -> float:
        """deterministic shift fitting the market curve: psi(t) = lambda_mkt(t) + D(t) - y0 E(t)  (cirpp.py:137-142)"""
        return self._lambda_market(t) + self._D(t) - self._pf(3) * self._E(t)

    def lambda_t(self, t, y_t):
        """default intensity lambda(t) = y_t + psi(t)  (cirpp.py:144-147)"""
        return y_t + self.psi(float(t))

    def _cond_survival_coeffs(self, t, T) -> tuple[float, float]:
        """S(t,T | y) = c * exp(-B(t,T) y)  (cirpp.py:246-285); returns (c, B)"""
        t, T = float(t), float(T)
        y0 = self._pf(3)
        pref = (self._market_survival_probability(T) / self._market_survival_probability(t)) \
            * (self._A(0.0, t) / self._A(0.0, T)) * cexp(-self._B(0.0, t) * y0 + self._B(0.0, T) * y0)
        return pref * self._A(t, T), self._B(t, T)

    def survival_probability(self, t, T, y_t):
        y_t = torch.as_tensor(y_t, dtype=FLOAT, device=device)
        if self.deterministic:
            ratio = self._market_survival_probability(T) / self._market_survival_probability(t)
            return torch.ones_like(y_t) * ratio
        c, B = self._cond_survival_coeffs(t, T)
        return c * torch.exp(-B * y_t)

    # ---- native hooks -------------------------------------------------------------------------------------------
    def _slots(self):
        kind = _abi.MODEL_CIRPP_DET if self.deterministic else _abi.MODEL_CIRPP
        return [SlotSpec(kind, [self._pf(i) for i in range(4)], 2, 1)]

    def _initial_state(self):
        if self.deterministic:
            return [self._lambda_market(self.t0()), 0.0]               # cirpp.py:148-149
        return [self._pf(3), 0.0]

    def _step_aux(self, scheme, t1, dt):
        if self.deterministic:
            return [[self._lambda_market(t1), self._lambda_market(t1 + dt)]]   # cirpp.py:161-163
        if scheme != SimulationScheme.EULER:
            raise NotImplementedError("CIR++ is simulated with the Euler full-truncation scheme (cirpp.py:174-198); "
                                      "the reference's lognormal 'analytic' proxy is shape-inconsistent and unused.")
        return [[self.psi(t1)]]

    def _atom(self, req, asset_id):
        k = req.request_type
        if k == RT.CONDITIONAL_SURVIVAL_PROBABILITY:
            if self.deterministic:
                return AtomCoef(a=self._market_survival_probability(req.time2) / self._market_survival_probability(req.time1))
            c, B = self._cond_survival_coeffs(req.time1, req.time2)
            return AtomCoef(col=0, b=c, c0=0.0, c1=-B)
        if k == RT.SURVIVAL_PROBABILITY:
            return AtomCoef(col=1, b=1.0, c0=0.0, c1=-1.0)             # exp(-Lambda), cirpp.py:312-314
        raise NotImplementedError(f"Request type {k} not supported by CIRpp.")

    def _supports_scheme(self, scheme):
        return self.deterministic or scheme == SimulationScheme.EULER
