"""Schwartz two-factor commodity spot model around a baseline forward curve (reference: models/schwartz_two_factor.py:9-219):
log S(t) = log F0(t) + x(t) + y(t), x a mean-reverting short-term factor, y a drifting long-term Brownian factor, correlated rho.
params (gradient order) = [rate, short_term_mean_reversion, short_term_vol, long_term_drift, long_term_vol, rho];
state = [log S, x, y]; two normals per sub-step.  The path kernel keeps (x, y) in registers and writes log S = log F0(t) + x + y
at the stored dates (csrc/mcx_device.h, MCX_MODEL_S2F)."""
from __future__ import annotations

import math
from bisect import bisect_right

import torch

from .. import _abi
from ..common.enums import SimulationScheme
from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequestType as RT
from .model import AtomCoef, Model, SlotSpec


class SchwartzTwoFactorModel(Model):
    def __init__(self, calibration_date: float, curve_times: list[float], curve_values: list[float], rate: float,
                 short_term_mean_reversion: float, short_term_vol: float, long_term_drift: float, long_term_vol: float,
                 rho: float, asset_id: str | None = None):
        super().__init__(calibration_date=calibration_date, simulation_dim=2, state_dim=3,
                         asset_ids=[asset_id] if asset_id else None)
        if len(curve_times) != len(curve_values):
            raise ValueError("curve_times and curve_values must have identical lengths.")
        if len(curve_times) < 2:
            raise ValueError("At least two curve points are required.")
        if any(v <= 0.0 for v in curve_values):
            raise ValueError("Curve values must be strictly positive.")
        self.curve_times = [float(t) for t in curve_times]
        self.curve_values = torch.tensor(curve_values, dtype=FLOAT, device=device)
        self._curve = [float(v) for v in curve_values]
        self.model_params = [torch.tensor(v, dtype=FLOAT, device=device) for v in
                             (rate, short_term_mean_reversion, short_term_vol, long_term_drift, long_term_vol, rho)]

    def get_rate(self):
        return self.model_params[0]

    def get_short_term_mean_reversion(self):
        return self.model_params[1]

    def get_short_term_vol(self):
        return self.model_params[2]

    def get_long_term_drift(self):
        return self.model_params[3]

    def get_long_term_vol(self):
        return self.model_params[4]

    def get_rho(self):
        return self.model_params[5]

    def get_model_param_names(self) -> list[str]:
        return ["rate", "short_term_mean_reversion", "short_term_vol", "long_term_drift", "long_term_vol", "rho"]

    def _curve_value(self, time) -> float:
        """piecewise-linear baseline curve, flat outside its range (schwartz_two_factor.py:96-113)"""
        t = float(time)
        ts, vs = self.curve_times, self._curve
        if t <= ts[0]:
            return vs[0]
        if t >= ts[-1]:
            return vs[-1]
        hi = bisect_right(ts, t)
        lo = hi - 1
        return vs[lo] + (vs[hi] - vs[lo]) * ((t - ts[lo]) / (ts[hi] - ts[lo]))

    # ---- correlated increments (schwartz_two_factor.py:122-145) --------------------------------------------------
    def _get_correlation_matrix(self, simulation_scheme):
        rho = self._pf(5)
        return torch.tensor([[1.0, rho], [rho, 1.0]], dtype=FLOAT, device=device)

    def _get_covariance_matrix(self, delta_t):
        dt = float(delta_t)
        kappa, sig_s, sig_l, rho = self._pf(1), self._pf(2), self._pf(4), self._pf(5)
        var_s = sig_s * sig_s * dt if abs(kappa) <= 1e-12 else sig_s * sig_s * (1.0 - math.exp(-2.0 * kappa * dt)) / (2.0 * kappa)
        var_l = sig_l * sig_l * dt
        cov = rho * math.sqrt(max(var_s * var_l, 0.0))
        return torch.tensor([[var_s, cov], [cov, var_l]], dtype=FLOAT, device=device)

    # ---- native hooks -------------------------------------------------------------------------------------------
    def _slots(self):
        return [SlotSpec(_abi.MODEL_S2F, [self._pf(i) for i in range(6)] + [math.log(self._curve_value(self.t0()))], 3, 2)]

    def _initial_state(self):
        return [math.log(self._curve_value(self.t0())), 0.0, 0.0]

    def _step_aux(self, scheme, t1, dt):
        kappa = self._pf(1)
        decay = 1.0 if abs(kappa) <= 1e-12 else math.exp(-kappa * dt)
        return [[decay, math.log(self._curve_value(t1 + dt))]]

    def _atom(self, req, asset_id):
        k, r, t0 = req.request_type, self._pf(0), self.t0()
        if k == RT.SPOT:
            return AtomCoef(col=0, b=1.0, c0=0.0, c1=1.0)                        # exp(log S)
        if k == RT.DISCOUNT_FACTOR:
            return AtomCoef(a=math.exp(-r * (float(req.time1) - t0)))
        if k == RT.NUMERAIRE:
            return AtomCoef(a=math.exp(r * (float(req.time1) - t0)))
        if k == RT.FORWARD_RATE:
            return AtomCoef(a=math.exp(r * (float(req.time2) - float(req.time1))))
        if k == RT.LIBOR_RATE:
            tau = float(req.time2) - float(req.time1)
            return AtomCoef(a=(math.exp(r * tau) - 1.0) / tau)
        raise NotImplementedError(f"Request type {k} not supported.")

    def _supports_scheme(self, scheme):
        return scheme in (SimulationScheme.ANALYTICAL, SimulationScheme.EULER)
