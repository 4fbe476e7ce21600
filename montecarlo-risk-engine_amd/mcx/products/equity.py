"""Equity underlying: its value is the SPOT request (reference: products/equity.py:7-40)."""
from __future__ import annotations

from collections import defaultdict

from ..request_interface.request_types import AtomicRequest, AtomicRequestType
from .product import Product


class Equity(Product):
    def __init__(self, asset_id: str | None = None):
        super().__init__(asset_ids=[asset_id])
        self.composite_req_handle = None
        self.spot_requests = {(0, self.get_asset_id()): AtomicRequest(AtomicRequestType.SPOT)}

    def __eq__(self, other):
        return isinstance(other, Equity) and self.get_asset_id() == other.get_asset_id()

    def __hash__(self):
        return hash(self.get_asset_id())

    def get_atomic_requests_for_underlying(self):
        out = defaultdict(list)
        for label, req in self.spot_requests.items():
            out[label].append(req)
        return out

    def _observed_from(self, observation_date):
        return Equity(self.get_asset_id())

    def _value_terms(self, ctx, time):
        return [(1.0, ctx.atom(AtomicRequest(AtomicRequestType.SPOT), self.get_asset_id(), time))]

    def get_value(self, resolved_atomic_requests):
        return resolved_atomic_requests[self.spot_requests[(0, self.get_asset_id())].handle]
