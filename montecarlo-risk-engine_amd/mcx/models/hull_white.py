"""Hull-White one-factor short-rate model  dr = (theta(t) - a r) dt + sigma dW  with theta(t) fitted to an initial
forward curve.

Reference status: `models/hull_white.py` is an unfinished draft (`# TODO: Fix!!`, `from model import *`, its own path
loops not wired to the engine, wrong `resolve_request` arity) — it cannot be imported, no reference test touches it, so
there is NO oracle for this model: **parity UNPINNED**.  This class keeps the draft's constructor arguments, parameter
order and its theta(t) (hull_white.py:58-63) and exact/Euler recursions (:76-88, :104-110), but plugs into the engine like
every other model (state [r, log B], left-endpoint accumulation of log B as in Vasicek, vasicek.py:80) and prices zero
bonds with the standard HW1F closed form  P(t,T|r) = P(0,T)/P(0,t) exp(B f(0,t) - sigma^2/(4a) (1-e^{-2at}) B^2 - B r).
Anchors used instead of an oracle (tests/test_hull_white.py): the constant-theta special case reproduces Vasicek paths,
and E[exp(-int r)] reprices the input curve."""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import _abi
from ..common.enums import SimulationScheme
from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequestType as RT
from .model import AtomCoef, Model, SlotSpec


class HullWhiteModel(Model):
    def __init__(self, calibration_date: float, rate: float, initial_forward_curve, forward_curve_derivative,
                 mean_reversion: float, volatility: float, asset_id: str | None = None, curve_times=None):
        super().__init__(calibration_date=calibration_date, state_dim=2, asset_ids=[asset_id])
        self._f = [float(v) for v in initial_forward_curve]
        self._df = [float(v) for v in forward_curve_derivative]
        assert len(self._f) == len(self._df) >= 2
        # the draft spreads the curve nodes over [0, 1] (hull_white.py:26); pass curve_times for a real tenor grid
        self._t = [float(v) for v in (np.linspace(0.0, 1.0, len(self._f)) if curve_times is None else curve_times)]
        self.model_params = [torch.tensor(v, dtype=FLOAT, device=device)
                             for v in (rate, volatility, mean_reversion, *self._f, *self._df)]

    def get_model_param_names(self) -> list[str]:
        n = len(self._f)
        return ["rate", "sigma", "mean_reversion", *[f"forward_curve[{i}]" for i in range(n)],
                *[f"forward_curve_derivative[{i}]" for i in range(n)]]

    def get_rate(self):
        return torch.stack([self.model_params[0]])

    def get_volatility(self):
        return torch.stack([self.model_params[1]])

    def get_mean_reversion_speed(self):
        return torch.stack([self.model_params[2]])

    # ---- curve helpers ------------------------------------------------------------------------------------------
    def interpolate(self, t: float, curve) -> float:
        """linear interpolation, last segment extrapolated (hull_white.py:41-49)"""
        ts = self._t
        idx = int(np.searchsorted(ts, t)) - 1
        idx = min(max(idx, 0), len(ts) - 2)
        return curve[idx] + (curve[idx + 1] - curve[idx]) * (t - ts[idx]) / (ts[idx + 1] - ts[idx])

    def compute_theta(self, t: float) -> float:
        a, sigma = self._pf(2), self._pf(1)
        return self.interpolate(t, self._df) + a * self.interpolate(t, self._f) \
            + (sigma ** 2 / (2 * a)) * (1 - math.exp(-2 * a * t))

    def _int_forward(self, t: float) -> float:
        """int_0^t f(0,s) ds of the piecewise-linear (extrapolated) forward curve, exact"""
        ts = self._t
        knots = [0.0] + [x for x in ts if 0.0 < x < t] + [t]
        total = 0.0
        for lo, hi in zip(knots[:-1], knots[1:]):
            total += 0.5 * (self.interpolate(lo, self._f) + self.interpolate(hi, self._f)) * (hi - lo)
        return total

    def discount_curve(self, t: float) -> float:
        return math.exp(-self._int_forward(t - self.t0()))

    def _zcb_coeffs(self, time1: float, time2: float) -> tuple[float, float]:
        a, sigma = self._pf(2), self._pf(1)
        t = time1 - self.t0()
        tau = time2 - time1
        B = (1 - math.exp(-a * tau)) / a
        alpha = math.log(self.discount_curve(time2) / self.discount_curve(time1)) + B * self.interpolate(t, self._f) \
            - (sigma ** 2 / (4 * a)) * (1 - math.exp(-2 * a * t)) * B ** 2
        return alpha, B

    def compute_bond_price(self, time1, time2, rate):
        alpha, B = self._zcb_coeffs(float(time1), float(time2))
        return math.exp(alpha) * torch.exp(-B * torch.as_tensor(rate, dtype=FLOAT, device=device))

    def _get_covariance_matrix(self, delta_t) -> torch.Tensor:
        a, sigma = self._pf(2), self._pf(1)
        return torch.tensor([[(sigma ** 2 / (2 * a)) * (1 - math.exp(-2 * a * float(delta_t)))]], dtype=FLOAT)

    # ---- native hooks -------------------------------------------------------------------------------------------
    def _slots(self):
        return [SlotSpec(_abi.MODEL_HW, [self._pf(0), self._pf(1), 0.0, self._pf(2)], 2, 1)]

    def _initial_state(self):
        return [self._pf(0), 0.0]

    def _step_aux(self, scheme, t1, dt):
        a = self._pf(2)
        theta = self.compute_theta(t1 - self.t0())
        if scheme == SimulationScheme.ANALYTICAL:
            E = math.exp(-a * dt)
            return [[E, theta * (1 - E) / a]]          # r' = r E + theta(t1)(1-E)/a + w      hull_white.py:76-83
        return [[theta]]                               # r' = r + (theta(t1) - a r) dt + ...  hull_white.py:104-108

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
        if k == RT.LIBOR_RATE:
            alpha, B = self._zcb_coeffs(req.time1, req.time2)
            tau = req.time2 - req.time1
            return AtomCoef(col=0, a=-1.0 / tau, b=1.0 / tau, c0=-alpha, c1=B)
        if k == RT.NUMERAIRE:
            return AtomCoef(col=1, b=1.0, c0=0.0, c1=1.0)
        raise NotImplementedError(f"Request type {k} not supported by Hull-White.")

    def _supports_scheme(self, scheme):
        return scheme in (SimulationScheme.ANALYTICAL, SimulationScheme.EULER)
