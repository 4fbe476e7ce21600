"""Plain-vanilla interest-rate swap = floating leg - fixed leg for a payer (reference: products/swap.py:11-172)."""
from __future__ import annotations

from collections import defaultdict
from enum import Enum

import torch

from .. import _abi
from ..common.packages import FLOAT, device
from .bond import Bond
from .product import CashEvent, Product


class IRSType(Enum):
    PAYER = 0
    RECEIVER = 1


class InterestRateSwap(Product):
    def __init__(self, startdate: float, enddate: float, notional: float, fixed_rate: float, tenor_fixed: float,
                 tenor_float: float, irs_type: IRSType, asset_id: str | None = None):
        super().__init__(asset_ids=[asset_id])
        self.startdate, self.enddate, self.notional = startdate, enddate, notional
        self.fixed_rate, self.tenor_fixed, self.tenor_float, self.irs_type = fixed_rate, tenor_fixed, tenor_float, irs_type
        self.composite_req_handle = None
        self.fixed_leg = Bond(startdate, enddate, notional, tenor_fixed, pays_notional=False, fixed_rate=fixed_rate,
                              asset_id=asset_id)
        self.floating_leg = Bond(startdate, enddate, notional, tenor_float, pays_notional=False, asset_id=asset_id)
        self._dates = sorted(set(self.fixed_leg._dates) | set(self.floating_leg._dates))
        self.product_timeline = torch.tensor(self._dates, dtype=FLOAT)
        self.modeling_timeline = self.product_timeline
        self.regression_timeline = torch.tensor([], dtype=FLOAT, device=device)

    def _key(self):
        return (self.startdate, self.enddate, self.notional, self.fixed_rate, self.tenor_fixed, self.tenor_float)

    def __eq__(self, other):
        return isinstance(other, InterestRateSwap) and self._key() == other._key()

    def __hash__(self):
        return hash(self._key())

    def get_atomic_requests(self):
        out = defaultdict(list)
        for leg in (self.fixed_leg, self.floating_leg):
            for label, reqs in leg.get_atomic_requests().items():
                out[label].extend(reqs)
        return out

    def get_atomic_requests_for_underlying(self):
        out = defaultdict(list)
        for leg in (self.fixed_leg, self.floating_leg):
            for label, reqs in leg.get_atomic_requests_for_underlying().items():
                out[label].extend(reqs)
        return out

    def _observed_from(self, observation_date):
        return InterestRateSwap(observation_date, self.enddate, self.notional, self.fixed_rate, self.tenor_fixed,
                                self.tenor_float, self.irs_type, asset_id=self.get_asset_id())

    def _signs(self):
        return (1.0, -1.0) if self.irs_type == IRSType.PAYER else (-1.0, 1.0)   # (float, fixed)

    def _value_terms(self, ctx, time):
        sf, sx = self._signs()
        terms = [(sf * w, a) for w, a in self.floating_leg._value_terms(ctx, time)]
        terms += [(sx * w, a) for w, a in self.fixed_leg._value_terms(ctx, time)]
        return terms

    def _cash_events(self, ctx):
        """Reference quirk kept (request_interface.py:61-66 + swap.py:86-100): the legs' requests are labelled with the
        LEG-local payment index, but the label is looked up in the swap's merged timeline; with tenor_fixed !=
        tenor_float the coarser leg's numeraire / LIBOR are therefore read from the state at swap date #i instead of
        the leg's own payment date #i.  Such terms carry their own denominator atom."""
        from ..request_interface.request_types import AtomicRequest, AtomicRequestType
        sf, sx = self._signs()
        out = []
        for t in self._dates:
            ev_num = ctx.atom(AtomicRequest(AtomicRequestType.NUMERAIRE, t), "numeraire", t)
            terms = []
            for leg, sign in ((self.floating_leg, sf), (self.fixed_leg, sx)):
                if t in leg._dates:
                    i = leg._dates.index(t)
                    leg_terms, num = leg._leg_terms(ctx, i, observed_at=self._dates[i])
                    terms += [(sign * w, a, -1 if num == ev_num else num) for w, a in leg_terms]
            out.append(CashEvent(_abi.EV_CASHFLOW, t, terms))
        return out
