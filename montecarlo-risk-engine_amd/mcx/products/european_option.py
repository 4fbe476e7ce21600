"""European option on an equity, bond or swap underlying (reference: products/european_option.py:15-145).
Monte-Carlo payoff runs on the GPU (MCX_EV_OPTION); the Black-Scholes closed forms are host-side anchors."""
from __future__ import annotations

import math

import torch

from .. import _abi
from ..common.packages import FLOAT, device
from ..models.black_scholes import BlackScholesModel
from ..request_interface.request_types import AtomicRequest, AtomicRequestType
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

    # ---- Black-Scholes closed forms (european_option.py:88-145) ------------------------------------------------
    def _bs_price(self, spot: float, rate: float, sigma: float, tau: float) -> float:
        d1 = (math.log(spot / self._K) + (rate + 0.5 * sigma ** 2) * tau) / (sigma * math.sqrt(tau))
        d2 = d1 - sigma * math.sqrt(tau)
        if self.option_type == OptionType.CALL:
            return spot * _norm_cdf(d1) - self._K * math.exp(-rate * tau) * _norm_cdf(d2)
        return self._K * math.exp(-rate * tau) * _norm_cdf(-d2) - spot * _norm_cdf(-d1)

    def compute_pv_analytically(self, model: BlackScholesModel):
        return torch.tensor([self._bs_price(model._pf(0), model._pf(2), model._pf(1), self._T)], dtype=FLOAT)

    def compute_pv_analytically_torch(self, model, params):
        """the same closed form on torch scalars (`params` in model.get_model_params() order): the analytic-evaluation path
        differentiates it with torch.autograd exactly like the reference (controller.py:609-648, european_option.py:107-121)"""
        spot, sigma, rate = params[0], params[1], params[2]
        T, K = self._T, self._K
        sq = math.sqrt(T)
        d1 = (torch.log(spot / K) + (rate + 0.5 * sigma ** 2) * T) / (sigma * sq)
        d2 = d1 - sigma * sq
        cdf = lambda x: 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))
        if self.option_type == OptionType.CALL:
            return spot * cdf(d1) - K * torch.exp(-rate * T) * cdf(d2)
        return K * torch.exp(-rate * T) * cdf(-d2) - spot * cdf(-d1)

    def _d1(self, model):
        spot, sigma, rate = model._pf(0), model._pf(1), model._pf(2)
        return (math.log(spot / self._K) + (rate + 0.5 * sigma ** 2) * self._T) / (sigma * math.sqrt(self._T))

    def compute_dVegadSigma_analytically(self, model: BlackScholesModel):           # vomma, european_option.py:290-304
        d1 = self._d1(model)
        d2 = d1 - model._pf(1) * math.sqrt(self._T)
        pdf = math.exp(-0.5 * d1 * d1) / math.sqrt(2.0 * math.pi)
        return torch.tensor([model._pf(0) * pdf * math.sqrt(self._T) * d1 * d2 / model._pf(1)], dtype=FLOAT)

    def compute_dDeltadSpot_analytically(self, model: BlackScholesModel):           # gamma, european_option.py:306-320
        d1 = self._d1(model)
        pdf = math.exp(-0.5 * d1 * d1) / math.sqrt(2.0 * math.pi)
        return torch.tensor([pdf / (model._pf(0) * model._pf(1) * math.sqrt(self._T))], dtype=FLOAT)

    def supports_analytic_pv(self, model) -> bool:
        return isinstance(model, BlackScholesModel)

    def supports_analytic_exposure(self, model) -> bool:
        return isinstance(model, BlackScholesModel)
