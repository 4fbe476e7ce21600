"""FlexiCall: k exercise rights on a strip of European options with distinct exercise dates (reference:
products/flexicall.py:4-186).  A (k+1)-state exercise machine: at each date the holder exercises iff
immediate + continuation(state - 1) > continuation(state).  GPU: MCX_EV_EXERCISE events with aux[0] = 1 (include/mcx.h)."""
from __future__ import annotations

import torch

from .. import _abi
from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequest, AtomicRequestType
from .european_option import EuropeanOption
from .product import CashEvent, OptionType, Product, ProductFamily


class FlexiCall(Product):
    def __init__(self, underlyings: list[EuropeanOption], num_exercise_rights: int, asset_id: str | None = None):
        super().__init__(asset_ids=[asset_id], product_family=ProductFamily.FLEXICALL_EXERCISE)
        assert num_exercise_rights <= len(underlyings), "Number of exercise rights cannot exceed number of underlyings"
        assert all(o.option_type == underlyings[0].option_type for o in underlyings), \
            "All underlyings must have the same option type"
        self.underlyings = sorted(underlyings, key=lambda o: float(o.exercise_date[0]))
        assert all(float(self.underlyings[i].exercise_date[0]) < float(self.underlyings[i + 1].exercise_date[0])
                   for i in range(len(underlyings) - 1)), "Exercise dates must be distinct"
        if num_exercise_rights + 1 > _abi.MAX_STATES:
            raise ValueError(f"at most {_abi.MAX_STATES - 1} exercise rights (MCX_MAX_STATES)")
        self._dates = [float(o.exercise_date[0]) for o in self.underlyings]
        self.product_timeline = torch.tensor(self._dates, dtype=FLOAT, device=device)
        self.modeling_timeline = self.product_timeline
        self.regression_timeline = self.product_timeline
        self.num_exercise_rights = int(num_exercise_rights)
        aid = self.get_asset_id()
        self.numeraire_requests = {i: AtomicRequest(AtomicRequestType.NUMERAIRE, t) for i, t in enumerate(self._dates)}
        self.spot_requests = {(i, aid): AtomicRequest(AtomicRequestType.SPOT) for i in range(len(self._dates))}
        self.underlying_requests = {i: o.underlying_requests[0] for i, o in enumerate(self.underlyings)}

    def get_num_states(self):
        return self.num_exercise_rights + 1

    def get_initial_state(self):
        return self.num_exercise_rights

    def _cash_events(self, ctx):
        sign = 1.0 if self.underlyings[0].option_type == OptionType.CALL else -1.0
        last = len(self._dates) - 1
        out = []
        for i, (t, opt) in enumerate(zip(self._dates, self.underlyings)):
            und = self.underlying_requests[i].underlying_asset
            out.append(CashEvent(_abi.EV_EXERCISE, t, und._value_terms(ctx, t), strike=float(opt.strike[0]), sign=sign,
                                 x_asset=self.asset_ids[0], reg_idx=None if i == last else i, aux=(1.0, 0.0, 0.0, 0.0)))
        return out
