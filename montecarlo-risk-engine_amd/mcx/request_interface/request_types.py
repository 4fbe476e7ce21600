"""What products and metrics may ask of a model at a timeline date, and the handle bookkeeping around it.
API surface of the reference's request_interface/request_types.py (names, constructor argument order, hash / equality
semantics); the per-path resolution itself happens inside the HIP kernels as "atoms" (include/mcx.h)."""
from __future__ import annotations

from enum import Enum, unique


@unique
class AtomicRequestType(Enum):
    SPOT, DISCOUNT_FACTOR, NUMERAIRE, FORWARD_RATE, LIBOR_RATE, SURVIVAL_PROBABILITY, CONDITIONAL_SURVIVAL_PROBABILITY = range(1, 8)


class _KeyedRequest:
    """two requests are the same request iff their keys agree: de-duplication in dicts / sets relies on it"""
    __slots__ = ()

    def key(self):
        raise NotImplementedError

    def __hash__(self):
        return hash(self.key())

    def __eq__(self, other):
        return isinstance(other, _KeyedRequest) and self.key() == other.key()


class AtomicRequest(_KeyedRequest):
    """one per-path market quantity (type, optional id, up to two times); `handle` is the integer slot assigned by the
    request interface"""
    __slots__ = ("request_type", "time1", "time2", "id", "handle")

    def __init__(self, request_type: AtomicRequestType, time1=None, time2=None, id=None):
        self.request_type, self.time1, self.time2, self.id = request_type, time1, time2, id
        self.handle = None

    def key(self):
        return self.request_type, self.id, self.time1, self.time2

    def set_handle(self, idx: int) -> None:
        self.handle = idx

    def __repr__(self):
        return f"AtomicRequest({self.request_type.name}, t1={self.time1}, t2={self.time2}, id={self.id})"


class UnderlyingRequest(_KeyedRequest):
    """the value of an underlying product observed at some date; keyed on the (hashable) underlying itself, the handle
    lives on the underlying"""
    __slots__ = ("underlying_asset",)

    def __init__(self, underlying_asset):
        self.underlying_asset = underlying_asset

    def key(self):
        return self.underlying_asset

    def get_handle(self):
        return self.underlying_asset.composite_req_handle

    def set_handle(self, idx: int):
        self.underlying_asset.composite_req_handle = idx

    def get_atomic_requests(self):
        return self.underlying_asset.get_atomic_requests_for_underlying()

    def get_value(self, resolved_atomic_requests):
        return self.underlying_asset.get_value(resolved_atomic_requests)
