"""Bermudan / American option: a two-state exercise machine driven by Longstaff-Schwartz continuation values
(reference: products/bermudan_option.py:6-193). The exercise step itself is MCX_EV_EXERCISE on the GPU."""
from __future__ import annotations

import numpy as np
import torch

from .. import _abi
from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequest, AtomicRequestType
from .product import CashEvent, OptionType, Product, ProductFamily


class BermudanOption(Product):
    def __init__(self, underlying: Product, exercise_dates, strike: float, option_type: OptionType,
                 asset_id: str | None = None):
        super().__init__(asset_ids=[asset_id], product_family=ProductFamily.BERMUDAN_EXERCISE)
        self._K = float(strike)
        self.strike = torch.tensor([self._K], dtype=FLOAT, device=device)
        self.option_type = option_type
        self.product_timeline = torch.tensor(np.asarray(exercise_dates, dtype=np.float64), dtype=FLOAT, device=device)
        self.modeling_timeline = self.product_timeline
        self.regression_timeline = self.product_timeline
        self.num_exercise_rights = 1
        self._dates = [float(t) for t in self.product_timeline]
        self.numeraire_requests = {i: AtomicRequest(AtomicRequestType.NUMERAIRE, t) for i, t in enumerate(self._dates)}
        aid = self.asset_ids[0]
        self.spot_requests = {(i, aid): AtomicRequest(AtomicRequestType.SPOT) for i in range(len(self._dates))}
        for i, t in enumerate(self._dates):
            self.underlying_requests[i] = underlying.generate_underlying_requests_for_date(t)

    def get_num_states(self):
        return 2

    def get_initial_state(self):
        return 1

    def _cash_events(self, ctx):
        sign = 1.0 if self.option_type == OptionType.CALL else -1.0
        last = len(self._dates) - 1
        out = []
        for i, t in enumerate(self._dates):
            und = self.underlying_requests[i].underlying_asset
            out.append(CashEvent(_abi.EV_EXERCISE, t, und._value_terms(ctx, t), strike=self._K, sign=sign,
                                 x_asset=self.asset_ids[0], reg_idx=None if i == last else i))
        return out


class AmericanOption(BermudanOption):
    def __init__(self, underlying, maturity, num_exercise_dates, strike, option_type, asset_id: str | None = None):
        dates = np.linspace(0.0, maturity, num_exercise_dates) if num_exercise_dates > 1 else [maturity]
        super().__init__(underlying=underlying, exercise_dates=dates, strike=strike, option_type=option_type,
                         asset_id=asset_id)
