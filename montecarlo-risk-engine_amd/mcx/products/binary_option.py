"""Binary (digital) option: payment * fuzzy indicator of S_T above / below the strike (reference:
products/binary_option.py:6-64; the indicator is ALWAYS the linear ramp of maths.py:3-9 with eps = 1, fuzzy = True).
GPU: one MCX_EV_OPTION event in aggregation mode 3 (include/mcx.h)."""
from __future__ import annotations

import math

import torch

from .. import _abi
from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequest, AtomicRequestType
from .product import CashEvent, OptionType, Product, ProductFamily


class BinaryOption(Product):
    def __init__(self, maturity: float, strike: float, payment_amount: float, option_type: OptionType,
                 asset_id: str | None = None):
        super().__init__(asset_ids=[asset_id], product_family=ProductFamily.BINARY_TERMINAL_PAYOFF)
        self._T, self._K, self._amount = float(maturity), float(strike), float(payment_amount)
        self.maturity = torch.tensor([self._T], dtype=FLOAT, device=device)
        self.strike = torch.tensor([self._K], dtype=FLOAT, device=device)
        self.payment_amount = torch.tensor([self._amount], dtype=FLOAT, device=device)
        self.option_type = option_type
        self.product_timeline = torch.tensor([self._T], dtype=FLOAT, device=device)
        self.modeling_timeline = self.product_timeline
        self.regression_timeline = torch.tensor([], dtype=FLOAT, device=device)
        self.numeraire_requests = {0: AtomicRequest(AtomicRequestType.NUMERAIRE, self._T)}
        self.spot_requests = {(0, self.get_asset_id()): AtomicRequest(AtomicRequestType.SPOT)}

    def _cash_events(self, ctx):
        terms = [(1.0, ctx.atom(AtomicRequest(AtomicRequestType.SPOT), self.get_asset_id(), self._T))]
        sign = 1.0 if self.option_type == OptionType.CALL else -1.0
        return [CashEvent(_abi.EV_OPTION, self._T, terms, strike=self._K, sign=sign, aux=(3.0, self._amount, 1.0, 0.0))]

    def payoff(self, spots, model):
        dot = torch.clamp((spots - self.strike + 1.0) / 2.0, 0.0, 1.0)
        return self.payment_amount * (dot if self.option_type == OptionType.CALL else 1.0 - dot)

    def compute_pv_analytically(self, model) -> torch.Tensor:                      # binary_option.py:46-56
        spot, sigma, rate = model._pf(0), model._pf(1), model._pf(2)
        d2 = (math.log(spot / self._K) + (rate - 0.5 * sigma ** 2) * self._T) / (sigma * math.sqrt(self._T))
        cdf = 0.5 * (1.0 + math.erf((d2 if self.option_type == OptionType.CALL else -d2) / math.sqrt(2.0)))
        return torch.tensor([self._amount * math.exp(-rate * self._T) * cdf], dtype=FLOAT)
