"""Asian option: payoff on the arithmetic or geometric average of the spot over `num_observation_timepoints` equidistant
dates (reference: products/asian_option.py:11-95).

GPU: ONE MCX_EV_OPTION event at maturity whose value terms are (1/n) * SPOT at each observation date — an atom carries its
own date index, so the path dependence needs no running state — in aggregation mode 0 (arithmetic) or 1 (geometric,
exp(mean log(S + 1e-10))).  Reference quirk reproduced: the payoff is divided by `numeraire_requests[len(product_timeline)
- 1]` = the numeraire of the FIRST observation date (asian_option.py:88), not of the maturity."""
from __future__ import annotations

from enum import Enum

import torch

from .. import _abi
from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequest, AtomicRequestType
from .product import CashEvent, OptionType, Product, ProductFamily


class AsianAveragingType(Enum):
    ARITHMETIC = 0
    GEOMETRIC = 1


class AsianOption(Product):
    def __init__(self, startdate: float, maturity: float, strike: float, num_observation_timepoints: int,
                 option_type: OptionType, averaging_type: AsianAveragingType = AsianAveragingType.ARITHMETIC,
                 asset_id: str | None = None):
        super().__init__(asset_ids=[asset_id], product_family=ProductFamily.ASIAN_PATH_TERMINAL)
        self._T, self._K = float(maturity), float(strike)
        self.maturity = torch.tensor([self._T], dtype=FLOAT, device=device)
        self.strike = torch.tensor([self._K], dtype=FLOAT, device=device)
        self.option_type = option_type
        self.averaging_type = averaging_type
        self.product_timeline = torch.tensor([self._T], dtype=FLOAT, device=device)
        self.modeling_timeline = torch.linspace(startdate, maturity, num_observation_timepoints, dtype=FLOAT, device=device)
        self.regression_timeline = torch.tensor([], dtype=FLOAT, device=device)
        self.numeraire_requests = {i: AtomicRequest(AtomicRequestType.NUMERAIRE, float(t))
                                   for i, t in enumerate(self.modeling_timeline)}
        self.spot_requests = {(i, self.get_asset_id()): AtomicRequest(AtomicRequestType.SPOT)
                              for i in range(len(self.modeling_timeline))}

    def _cash_events(self, ctx):
        obs = [float(t) for t in self.modeling_timeline]
        w = 1.0 / len(obs)
        terms = [(w, ctx.atom(AtomicRequest(AtomicRequestType.SPOT), self.get_asset_id(), t)) for t in obs]
        sign = 1.0 if self.option_type == OptionType.CALL else -1.0
        mode = 1.0 if self.averaging_type == AsianAveragingType.GEOMETRIC else 0.0
        return [CashEvent(_abi.EV_OPTION, self._T, terms, strike=self._K, sign=sign, aux=(mode, 0.0, 0.0, 0.0),
                          num_time=obs[len(self.product_timeline) - 1])]

    @staticmethod
    def _average_paths(spots: torch.Tensor, averaging_type: AsianAveragingType) -> torch.Tensor:
        if averaging_type == AsianAveragingType.GEOMETRIC:
            return torch.exp(torch.mean(torch.log(spots + 1e-10), dim=1))
        return torch.mean(spots, dim=1)

    def payoff(self, spots, model):
        avg = AsianOption._average_paths(spots, self.averaging_type)
        sign = 1.0 if self.option_type == OptionType.CALL else -1.0
        return torch.clamp(sign * (avg - self.strike), min=0.0)

    def compute_pv_analytically(self, model):
        raise NotImplementedError("Analytical Asian pricing is not implemented for this product.")
