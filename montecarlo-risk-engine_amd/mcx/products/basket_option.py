"""Basket option: arithmetic or geometric weighted basket payoff at maturity, optionally with the geometric basket as
control variate (reference: products/basket_option.py:10-142).  The payoff is ONE MCX_EV_OPTION event over the weighted
SPOT atoms of the basket's assets; its aggregation mode travels in the event's aux[0] (include/mcx.h)."""
from __future__ import annotations

import math
from enum import Enum

import torch

from .. import _abi
from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequest, AtomicRequestType
from .product import CashEvent, OptionType, Product, ProductFamily


class BasketOptionType(Enum):
    ARITHMETIC = 0
    GEOMETRIC = 1


def _norm_cdf(x: float) -> float:
    return 0.5 * (1.0 + math.erf(x / math.sqrt(2.0)))


class BasketOption(Product):
    def __init__(self, maturity: float, asset_ids: list[str], weights, strike: float, option_type: OptionType,
                 basket_option_type: BasketOptionType = BasketOptionType.ARITHMETIC, use_variation_reduction: bool = False):
        super().__init__(asset_ids=list(asset_ids), product_family=ProductFamily.BASKET_TERMINAL_PAYOFF)
        self._T, self._K = float(maturity), float(strike)
        self._w = [float(w) for w in weights]
        assert len(self._w) == len(self.asset_ids)
        self.maturity = torch.tensor([self._T], dtype=FLOAT, device=device)
        self.strike = torch.tensor([self._K], dtype=FLOAT, device=device)
        self.weights = torch.tensor(self._w, dtype=FLOAT, device=device)
        self.option_type = option_type
        self.product_timeline = torch.tensor([self._T], dtype=FLOAT, device=device)
        self.modeling_timeline = self.product_timeline
        self.regression_timeline = torch.tensor([], dtype=FLOAT, device=device)
        self.basket_option_type = basket_option_type
        self.use_variation_reduction = use_variation_reduction
        self.numeraire_requests = {0: AtomicRequest(AtomicRequestType.NUMERAIRE, self._T)}
        self.spot_requests = {(0, a): AtomicRequest(AtomicRequestType.SPOT) for a in self.asset_ids}
        self._model = None

    def _sign(self) -> float:
        return 1.0 if self.option_type == OptionType.CALL else -1.0

    def _cash_events(self, ctx):
        terms = [(w, ctx.atom(AtomicRequest(AtomicRequestType.SPOT), a, self._T)) for w, a in zip(self._w, self.asset_ids)]
        if self.use_variation_reduction:
            # payoff_classical - payoff_geometric + analytic geometric PV, then divided by the numeraire like every payoff
            # (basket_option.py:75-82, 102-110)
            aux = (2.0 if self.basket_option_type == BasketOptionType.ARITHMETIC else 1.0,
                   float(self.compute_pv_analytically(ctx.model)), 0.0, 0.0)
            if self.basket_option_type == BasketOptionType.GEOMETRIC:          # geo - geo + correction
                return [CashEvent(_abi.EV_CASHFLOW, self._T, [(aux[1], ctx.const_atom(1.0))])]
        else:
            aux = (1.0 if self.basket_option_type == BasketOptionType.GEOMETRIC else 0.0, 0.0, 0.0, 0.0)
        return [CashEvent(_abi.EV_OPTION, self._T, terms, strike=self._K, sign=self._sign(), aux=aux)]

    # ---- host evaluation (API parity) ------------------------------------------------------------------------------
    def compute_payoff(self, spots: torch.Tensor, basket_option_type: BasketOptionType) -> torch.Tensor:
        zero = torch.tensor(0.0, dtype=FLOAT, device=spots.device)
        if basket_option_type == BasketOptionType.ARITHMETIC:
            basket = (spots * self.weights).sum(dim=1)
        else:
            basket = torch.exp((torch.log(spots + 1e-10) * self.weights).sum(dim=1))
        return torch.maximum(self._sign() * (basket - self.strike), zero)

    def payoff(self, spots, model):
        if self.use_variation_reduction:
            return (self.compute_payoff(spots, self.basket_option_type)
                    - self.compute_payoff(spots, BasketOptionType.GEOMETRIC) + self.compute_pv_analytically(model))
        return self.compute_payoff(spots, self.basket_option_type)

    def compute_pv_analytically(self, model) -> torch.Tensor:
        """geometric basket under (multi-asset) Black-Scholes, formulas exactly as basket_option.py:112-141 (equal-weight
        geometric mean of the spots, `model._get_covariance_matrix(T)` as the variance input)"""
        S = [float(v) for v in model.get_spot().detach().reshape(-1)]
        r = float(model.get_rate().detach().reshape(-1)[0])
        sig = [float(v) for v in model.get_volatility().detach().reshape(-1)]
        T, K, n = self._T, self._K, len(S)
        f_s_bar = math.exp(sum(math.log(s) for s in S) / n)
        cov = model._get_covariance_matrix(T).detach()
        w = self.weights.to(cov.dtype)
        sigma = math.sqrt(float(torch.dot(w, torch.mv(cov, w))))
        F = f_s_bar * math.exp((r - 0.5 * sum(s * s for s in sig) / n + 0.5 * sigma ** 2) * T)
        sst = sigma * math.sqrt(T)
        d1 = (math.log(F / K) + 0.5 * sigma ** 2 * T) / sst
        d2 = d1 - sst
        if self.option_type == OptionType.CALL:
            pv = math.exp(-r * T) * (F * _norm_cdf(d1) - K * _norm_cdf(d2))
        else:
            pv = math.exp(-r * T) * (K * _norm_cdf(-d2) - F * _norm_cdf(-d1))
        return torch.tensor([pv], dtype=FLOAT)
