"""Fixed / floating coupon bonds and zero bonds (reference: products/bond.py:6-214).

Schedule quirks kept on purpose (parity): dates are built by repeated `date += tenor` from startdate (bond.py:37-68);
coupons are NOT scaled by the notional (bond.py:180, 208) — only the redemption is; a floating coupon uses the LIBOR
for (t - tenor, t) evaluated from the state AT the payment date (request registered at the payment index, bond.py:56)."""
from __future__ import annotations

from collections import defaultdict

import torch

from .. import _abi
from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequest, AtomicRequestType as RT
from .product import CashEvent, Product


class Bond(Product):
    def __init__(self, startdate: float, maturity: float, notional: float, tenor: float, pays_notional=True,
                 fixed_rate=None, asset_id: str | None = None):
        super().__init__(asset_ids=[asset_id])
        self._start, self._mat, self._notional, self._tenor = float(startdate), float(maturity), float(notional), float(tenor)
        self._rate = None if fixed_rate is None else float(fixed_rate)
        self.startdate = torch.tensor([self._start], dtype=FLOAT, device=device)
        self.maturity = torch.tensor([self._mat], dtype=FLOAT, device=device)
        self.notional = torch.tensor([self._notional], dtype=FLOAT, device=device)
        self.tenor = torch.tensor([self._tenor], dtype=FLOAT, device=device)
        self.fixed_rate = None if fixed_rate is None else torch.tensor([self._rate], dtype=FLOAT, device=device)
        self.pays_notional = pays_notional
        self.composite_req_handle = None

        aid = self.get_asset_id()
        self.atomic_requests_for_underlying: dict = {}
        dates: list[float] = []
        self._accrual_starts: list[float] = []      # float leg: start of the LIBOR period of each payment
        date, idx = self._start + self._tenor, 0
        while date < self._mat:
            self.numeraire_requests[idx] = AtomicRequest(RT.NUMERAIRE, date)
            if self._rate is None:
                self.libor_requests[(idx, aid)] = AtomicRequest(RT.LIBOR_RATE, date - self._tenor, date)
                self.atomic_requests_for_underlying[(idx, aid)] = AtomicRequest(RT.FORWARD_RATE, self._start, date - self._tenor)
                self._accrual_starts.append(date - self._tenor)
            else:
                self.atomic_requests_for_underlying[(idx, aid)] = AtomicRequest(RT.FORWARD_RATE, self._start, date)
            dates.append(date)
            date += self._tenor
            idx += 1
        self.numeraire_requests[idx] = AtomicRequest(RT.NUMERAIRE, self._mat)
        if self._rate is None:
            self.libor_requests[(idx, aid)] = AtomicRequest(RT.LIBOR_RATE, date - self._tenor, self._mat)
            self.atomic_requests_for_underlying[(idx, aid)] = AtomicRequest(RT.FORWARD_RATE, self._start, date - self._tenor)
            self.atomic_requests_for_underlying[(idx + 1, aid)] = AtomicRequest(RT.FORWARD_RATE, self._start, self._mat)
            self._accrual_starts.append(date - self._tenor)
        else:
            self.atomic_requests_for_underlying[(idx, aid)] = AtomicRequest(RT.FORWARD_RATE, self._start, self._mat)
        dates.append(self._mat)
        self._dates = dates
        self.payment_dates = torch.tensor(dates, dtype=FLOAT, device=device)
        self.product_timeline = self.payment_dates
        self.modeling_timeline = self.payment_dates
        self.regression_timeline = torch.tensor([], dtype=FLOAT, device=device)

    def __eq__(self, other):
        return (isinstance(other, Bond) and self._start == other._start and self._mat == other._mat
                and self._tenor == other._tenor and self._rate == other._rate and self.pays_notional == other.pays_notional)

    def __hash__(self):
        return hash((self._start, self._mat, self._tenor, self._rate, self.pays_notional))

    def get_atomic_requests_for_underlying(self):
        out = defaultdict(list)
        for label, req in self.atomic_requests_for_underlying.items():
            out[label].append(req)
        return out

    def _observed_from(self, observation_date):
        return Bond(observation_date, self._mat, self._notional, self._tenor, self.pays_notional, self._rate,
                    asset_id=self.get_asset_id())

    def _period(self, i: int) -> float:
        prev = self._start if i == 0 else self._dates[i - 1]
        return self._dates[i] - prev

    # value observed at `time` (bond.py:115-163): discounted remaining cashflows off the ZCB curve P(start, .)
    def _value_terms(self, ctx, time):
        aid = self.get_asset_id()
        reqs = self.atomic_requests_for_underlying
        n = len(self._dates)
        terms = []
        if self._rate is not None:
            for i in range(n):
                terms.append((self._notional * self._rate * self._period(i), ctx.atom(reqs[(i, aid)], aid, time)))
        else:
            for i in range(n):
                terms.append((self._notional, ctx.atom(reqs[(i, aid)], aid, time)))
                terms.append((-self._notional, ctx.atom(reqs[(i + 1, aid)], aid, time)))
        if self.pays_notional:
            terms.append((self._notional, ctx.atom(reqs[(n - 1, aid)], aid, time)))
        return terms

    # per payment date (bond.py:171-214)
    def _leg_terms(self, ctx, i: int, observed_at: float | None = None) -> tuple[list, int]:
        """terms of payment i and the atom id of ITS numeraire; `observed_at` = date whose simulated state resolves the
        payment's requests (the payment date itself, except inside a swap with unequal leg tenors — see swap.py)"""
        aid = self.get_asset_id()
        t_obs = self._dates[i] if observed_at is None else observed_at
        dt = self._period(i)
        last = i == len(self._dates) - 1
        num = ctx.atom(self.numeraire_requests[i], "numeraire", t_obs)
        if self._rate is not None:
            cash = self._rate * dt + (self._notional if (self.pays_notional and last) else 0.0)
            return [(1.0, ctx.const_atom(cash))], num
        terms = [(dt, ctx.atom(self.libor_requests[(i, aid)], aid, t_obs))]
        if self.pays_notional and last:
            terms.append((1.0, ctx.const_atom(self._notional)))
        return terms, num

    def _cash_events(self, ctx):
        return [CashEvent(_abi.EV_CASHFLOW, t, self._leg_terms(ctx, i)[0]) for i, t in enumerate(self._dates)]
