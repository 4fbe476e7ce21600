"""European option on an equity, bond or swap underlying (reference: products/european_option.py:15-145).
Monte-Carlo payoff runs on the GPU (MCX_EV_OPTION); the Black-Scholes closed forms are host-side anchors."""
from __future__ import annotations

import math

import torch

from .. import _abi
from ..common.packages import FLOAT, device
from ..models.black_scholes import BlackScholesModel
from ..models.black_scholes_multi import BlackScholesMulti
from ..request_interface.request_types import AtomicRequest, AtomicRequestType
from .equity import Equity
from .product import CashEvent, OptionType, Product, ProductFamily


def _norm_cdf(x: float) -> float:
    return 0.5 * (1.0 + math.erf(x / math.sqrt(2.0)))


class EuropeanOption(Product):
    def __init__(self, underlying: Product, exercise_date: float, strike: float, option_type: OptionType,
                 asset_id: str | None = None):
        super().__init__(asset_ids=[asset_id], product_family=ProductFamily.VANILLA_TERMINAL_OPTION)
        self._T, self._K = float(exercise_date), float(strike)
        self.exercise_date = torch.tensor([self._T], dtype=FLOAT, device=device)
        self.strike = torch.tensor([self._K], dtype=FLOAT, device=device)
        self.option_type = option_type
        self.product_timeline = torch.tensor([self._T], dtype=FLOAT, device=device)
        self.modeling_timeline = self.product_timeline
        self.regression_timeline = torch.tensor([], dtype=FLOAT, device=device)
        self.underlying = underlying
        self.numeraire_requests = {0: AtomicRequest(AtomicRequestType.NUMERAIRE, self._T)}
        self.underlying_requests = {0: underlying.generate_underlying_requests_for_date(self._T)}

    def _sign(self) -> float:
        return 1.0 if self.option_type == OptionType.CALL else -1.0

    def _cash_events(self, ctx):
        und = self.underlying_requests[0].underlying_asset
        return [CashEvent(_abi.EV_OPTION, self._T, und._value_terms(ctx, self._T), strike=self._K, sign=self._sign())]

    def _cash_template_key(self):
        """options on the spot of one asset with one maturity differ in strike and sign only: the controller builds their cash event
        (atoms, terms) once per key and re-uses it (books of 10^4-10^5 Europeans: pv_performance_large_netting_set.py).  None: no
        shortcut (any other underlying carries its own parameters)."""
        und = self.underlying_requests[0].underlying_asset
        return (type(self).__name__, self._T, und.get_asset_id(), self.get_asset_id()) if type(und) is Equity else None

    def _cash_template_fill(self, row: tuple) -> tuple:
        return row[:8] + (float(self._K), float(self._sign())) + row[10:]

    # ---- Black-Scholes closed forms (european_option.py:88-145) ------------------------------------------------
    def _bs_price(self, spot: float, rate: float, sigma: float, tau: float) -> float:
        d1 = (math.log(spot / self._K) + (rate + 0.5 * sigma ** 2) * tau) / (sigma * math.sqrt(tau))
        d2 = d1 - sigma * math.sqrt(tau)
        if self.option_type == OptionType.CALL:
            return spot * _norm_cdf(d1) - self._K * math.exp(-rate * tau) * _norm_cdf(d2)
        return self._K * math.exp(-rate * tau) * _norm_cdf(-d2) - spot * _norm_cdf(-d1)

    def _bs_param_indices(self, model) -> tuple[int, int, int]:
        """positions of (spot, sigma, rate) of THIS option's asset in model.get_model_params()
        (european_option.py:70-86: a BlackScholesMulti is addressed through the option's asset id)"""
        if isinstance(model, BlackScholesMulti):
            asset_id = self.get_asset_id()
            if asset_id not in model.asset_ids:
                raise ValueError(f"Asset id '{asset_id}' not found in model asset ids {model.asset_ids}.")
            k, n = model.asset_ids.index(asset_id), model.num_assets
            return k, n + k, 2 * n
        return 0, 1, 2

    def _bs_inputs(self, model) -> tuple[float, float, float]:
        i_s, i_v, i_r = self._bs_param_indices(model)
        return model._pf(i_s), model._pf(i_v), model._pf(i_r)

    def compute_pv_analytically(self, model: BlackScholesModel | BlackScholesMulti):
        spot, sigma, rate = self._bs_inputs(model)
        return torch.tensor([self._bs_price(spot, rate, sigma, self._T)], dtype=FLOAT)

    def compute_pv_analytically_torch(self, model, params):
        """the same closed form on torch scalars (`params` in model.get_model_params() order): the analytic-evaluation path
        differentiates it with torch.autograd exactly like the reference (controller.py:609-648, european_option.py:107-121)"""
        i_s, i_v, i_r = self._bs_param_indices(model)
        spot, sigma, rate = params[i_s], params[i_v], params[i_r]
        T, K = self._T, self._K
        sq = math.sqrt(T)
        d1 = (torch.log(spot / K) + (rate + 0.5 * sigma ** 2) * T) / (sigma * sq)
        d2 = d1 - sigma * sq
        cdf = lambda x: 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))
        if self.option_type == OptionType.CALL:
            return spot * cdf(d1) - K * torch.exp(-rate * T) * cdf(d2)
        return K * torch.exp(-rate * T) * cdf(-d2) - spot * cdf(-d1)

    def _d1(self, model):
        spot, sigma, rate = self._bs_inputs(model)
        return (math.log(spot / self._K) + (rate + 0.5 * sigma ** 2) * self._T) / (sigma * math.sqrt(self._T))

    def compute_dVegadSigma_analytically(self, model: BlackScholesModel):           # vomma, european_option.py:290-304
        d1 = self._d1(model)
        spot, sigma, _ = self._bs_inputs(model)
        d2 = d1 - sigma * math.sqrt(self._T)
        pdf = math.exp(-0.5 * d1 * d1) / math.sqrt(2.0 * math.pi)
        return torch.tensor([spot * pdf * math.sqrt(self._T) * d1 * d2 / sigma], dtype=FLOAT)

    def compute_dDeltadSpot_analytically(self, model: BlackScholesModel):           # gamma, european_option.py:306-320
        d1 = self._d1(model)
        pdf = math.exp(-0.5 * d1 * d1) / math.sqrt(2.0 * math.pi)
        spot, sigma, _ = self._bs_inputs(model)
        return torch.tensor([pdf / (spot * sigma * math.sqrt(self._T))], dtype=FLOAT)

    def supports_analytic_pv(self, model) -> bool:
        return isinstance(model, (BlackScholesModel, BlackScholesMulti))

    def supports_analytic_exposure(self, model) -> bool:
        return isinstance(model, (BlackScholesModel, BlackScholesMulti))


    # ---- Heston semi-analytic call price (european_option.py:147-262): P1/P2 Fourier integrals of the characteristic function
    # in its numerically stable form (root of d with Re(d) <= 0), scipy.integrate.quad on [0, 100] like the reference
    def heston_cf(self, idx, u, T, S0, r, params):
        import numpy as np
        kappa, theta, sigma, rho, v0 = params
        b, shift = (kappa - rho * sigma, 0.5) if idx == 1 else (kappa, -0.5)
        if idx not in (1, 2):
            raise ValueError("idx must be 1 or 2")
        iu = 1j * u
        beta = b - rho * sigma * iu
        d = np.sqrt(beta ** 2 + sigma ** 2 * (u ** 2 - 2.0 * iu * shift))
        if np.real(d) > 0:
            d = -d
        g = (beta - d) / (beta + d)
        e = np.exp(-d * T)
        C = r * iu * T + (kappa * theta / sigma ** 2) * ((beta - d) * T - 2.0 * np.log((1.0 - g * e) / (1.0 - g)))
        D = ((beta - d) / sigma ** 2) * ((1.0 - e) / (1.0 - g * e))
        return np.exp(C + D * v0 + iu * np.log(S0))

    def _Qj(self, j, S0, K, T, r, params):
        import numpy as np
        from scipy.integrate import quad
        if j not in (1, 2):
            raise ValueError("j must be 1 or 2")
        f = lambda x: float(np.real(np.exp(-1j * x * np.log(K)) * self.heston_cf(j, x + 0j, T, S0, r, params) / (1j * x)))
        integral, _ = quad(f, 0.0, 100.0, limit=200)
        return 0.5 + integral / np.pi

    def heston_call_price(self, model, K: float, T: float):
        spot, sigma, rate, rho, kappa, theta, v0 = (model._pf(i) for i in range(7))
        params = (kappa, theta, sigma, rho, v0)
        return spot * self._Qj(1, spot, K, T, rate, params) - K * math.exp(-rate * T) * self._Qj(2, spot, K, T, rate, params)

    def compute_pv_bond_option_analytically(self, model):
        """European option on a zero-coupon bond under Vasicek (Jamshidian closed form, european_option.py:264-288)"""
        from .bond import Bond
        if not isinstance(self.underlying, Bond):
            raise TypeError("Expected self.underlying to be of type Bond")
        a, rate, sigma, t0 = model._pf(3), model._pf(0), model._pf(1), model.t0()
        T, S = self._T, float(self.underlying.maturity[0]) if hasattr(self.underlying.maturity, "__len__") else float(self.underlying.maturity)
        p_T = float(model.compute_bond_price(t0, T, torch.tensor([rate], dtype=FLOAT))[0])
        p_S = float(model.compute_bond_price(t0, S, torch.tensor([rate], dtype=FLOAT))[0])
        b_ts = (1.0 - math.exp(-a * (S - T))) / a
        sig = sigma * math.sqrt((1.0 - math.exp(-2.0 * a * (T - t0))) / (2.0 * a)) * b_ts
        if sig == 0.0:              # exercise at the bond's maturity: the closed form degenerates to the discounted intrinsic value
            call = max(p_S - self._K * p_T, 0.0)
            return torch.tensor([call if self.option_type == OptionType.CALL else call - (p_S - self._K * p_T)], dtype=FLOAT)
        d1 = (math.log(p_S / (p_T * self._K)) + 0.5 * sig * sig) / sig
        d2 = d1 - sig
        if self.option_type == OptionType.CALL:
            return torch.tensor([p_S * _norm_cdf(d1) - self._K * p_T * _norm_cdf(d2)], dtype=FLOAT)
        return torch.tensor([self._K * p_T * _norm_cdf(-d2) - p_S * _norm_cdf(-d1)], dtype=FLOAT)

    def compute_pv_analytically_heston(self, model):
        from ..models.heston import HestonModel
        if not isinstance(model, HestonModel):
            raise TypeError("Expected model to be of type HestonModel")
        return self.heston_call_price(model, self._K, self._T)
