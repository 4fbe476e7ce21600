"""Request vocabulary shared by products, metrics and models (reference: request_interface/request_types.py:10-68)."""
from __future__ import annotations

from enum import Enum


class AtomicRequestType(Enum):
    SPOT = 1
    DISCOUNT_FACTOR = 2
    NUMERAIRE = 3
    FORWARD_RATE = 4
    LIBOR_RATE = 5
    SURVIVAL_PROBABILITY = 6
    CONDITIONAL_SURVIVAL_PROBABILITY = 7


class AtomicRequest:
    """A per-path market quantity asked of the model at one timeline date. Hash/eq on (type, id, time1, time2)."""

    def __init__(self, request_type: AtomicRequestType, time1=None, time2=None, id=None):
        self.request_type = request_type
        self.id = id
        self.time1 = time1
        self.time2 = time2
        self.handle = None

    def set_handle(self, idx: int) -> None:
        self.handle = idx

    def key(self):
        return (self.request_type, self.id, self.time1, self.time2)

    def __eq__(self, other):
        return self.key() == other.key()

    def __hash__(self):
        return hash(self.key())


class UnderlyingRequest:
    """Composite request: the value of an underlying product observed at a date."""

    def __init__(self, underlying_asset):
        self.underlying_asset = underlying_asset

    def set_handle(self, idx: int):
        self.underlying_asset.composite_req_handle = idx

    def get_handle(self):
        return self.underlying_asset.composite_req_handle

    def get_atomic_requests(self):
        return self.underlying_asset.get_atomic_requests_for_underlying()

    def get_value(self, resolved_atomic_requests):
        return self.underlying_asset.get_value(resolved_atomic_requests)

    def key(self):
        return self.underlying_asset

    def __eq__(self, other):
        return self.key() == other.key()

    def __hash__(self):
        return hash(self.key())
