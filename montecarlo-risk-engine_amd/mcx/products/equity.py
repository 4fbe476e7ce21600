"""An equity as underlying of an option: one SPOT request; its value terms are that single atom with weight one."""
from __future__ import annotations

from collections import defaultdict

from ..request_interface.request_types import AtomicRequest, AtomicRequestType
from .product import Product

_SPOT = AtomicRequestType.SPOT


class Equity(Product):
    def __init__(self, asset_id: str | None = None):
        super().__init__(asset_ids=[asset_id])
        self.spot_requests = {(0, self.get_asset_id()): AtomicRequest(_SPOT)}
        self.composite_req_handle = None          # slot of the composite (underlying) request, set by the request interface

    # two Equity objects on the same asset are the same underlying (they key UnderlyingRequest)
    def __hash__(self):
        return hash(self.get_asset_id())

    def __eq__(self, other):
        return isinstance(other, Equity) and other.get_asset_id() == self.get_asset_id()

    def get_atomic_requests_for_underlying(self):
        grouped = defaultdict(list)
        for label, request in self.spot_requests.items():
            grouped[label].append(request)
        return grouped

    def get_value(self, resolved_atomic_requests):
        (request,) = self.spot_requests.values()
        return resolved_atomic_requests[request.handle]

    # ---- native hooks ------------------------------------------------------------------------------------------------
    def _value_terms(self, ctx, time):
        return [(1.0, ctx.atom(AtomicRequest(_SPOT), self.get_asset_id(), time))]

    def _observed_from(self, observation_date):
        return Equity(self.get_asset_id())
