"""Product base class (reference surface: products/product.py:13-228).

A product here is a *description*: its timelines, exercise-state space and — through `_cash_events` / `_value_terms` —
the event program the HIP evaluation kernel interprets per path (include/mcx.h "book program").  The per-path arithmetic
of the reference's compute_normalized_cashflows / get_value / continuation lookups runs on the GPU, not in Python."""
from __future__ import annotations

from collections import defaultdict
from dataclasses import dataclass, field
from enum import Enum

import torch

from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequest, AtomicRequestType, UnderlyingRequest


class OptionType(Enum):
    CALL = 1
    PUT = 2


class SettlementType(Enum):
    PHYSICAL = 0
    CASH = 1


class ProductFamily(Enum):
    GENERIC = "generic"
    VANILLA_TERMINAL_OPTION = "vanilla_terminal_option"
    BERMUDAN_EXERCISE = "bermudan_exercise"
    BASKET_TERMINAL_PAYOFF = "basket_terminal_payoff"
    BINARY_TERMINAL_PAYOFF = "binary_terminal_payoff"
    BARRIER_PATH_TERMINAL = "barrier_path_terminal"
    FLEXICALL_EXERCISE = "flexicall_exercise"
    ASIAN_PATH_TERMINAL = "asian_path_terminal"


@dataclass
class CashEvent:
    """one product-date event (kind = mcx MCX_EV_CASHFLOW / OPTION / EXERCISE)"""
    kind: int
    time: float
    terms: list                     # [(weight, atom_id)]
    strike: float = 0.0
    sign: float = 1.0
    x_asset: str | None = None      # EXERCISE: explanatory SPOT asset
    reg_idx: int | None = None      # EXERCISE: index into product.regression_coeffs (None: continuation 0)
    aux: tuple = (0.0, 0.0, 0.0, 0.0)   # OPTION: basket aggregation mode + control-variate constant (include/mcx.h)
    num_time: float | None = None       # date whose numeraire normalises the event (default: the event's own date)
    x_time: float | None = None         # date at which x_asset's spot is read (default: the event's own date)
    coeff_params: tuple = ()            # constants the event reads from the coefficient array at its coeff_off (bridge barrier)


class Product:
    def __init__(self, asset_ids: list[str] | None = None, product_id: int = 0,
                 product_family: ProductFamily = ProductFamily.GENERIC):
        self.asset_ids = asset_ids if asset_ids else [""]
        self.product_id = product_id
        self.name: str | None = None
        self.product_family = product_family
        self.spot_requests: dict = {}
        self.numeraire_requests: dict = {}
        self.libor_requests: dict = {}
        self.underlying_requests: dict = {}
        self.product_timeline: torch.Tensor | None = None
        self.modeling_timeline: torch.Tensor | None = None
        self.regression_timeline: torch.Tensor | None = None
        self.regression_coeffs: torch.Tensor | None = None

    # ---- reference API -------------------------------------------------------------------------------------------
    def get_atomic_requests(self):
        out = defaultdict(list)
        for t, req in self.numeraire_requests.items():
            out[(t, "numeraire")].append(req)
        for label, req in self.spot_requests.items():
            out[label].append(req)
        for label, req in self.libor_requests.items():
            out[label].append(req)
        return out

    def get_atomic_requests_for_underlying(self):
        return defaultdict(list)

    def get_underlying_requests(self):
        out = defaultdict(list)
        for t, req in self.underlying_requests.items():
            out[t].append(req)
        return out

    def get_num_states(self):
        return 1

    def get_state_dtype(self):
        return torch.long

    def get_initial_state(self):
        return 0

    def get_asset_id(self, id: int | None = None):
        return self.asset_ids[id] if id else self.asset_ids[0]

    def get_name(self) -> str:
        return self.name if self.name else self.__class__.__name__

    def get_product_family(self) -> ProductFamily:
        return self.product_family

    def _allocate_regression_coeffs(self, regression_function):
        self.regression_coeffs = torch.zeros(
            (len(self.regression_timeline), self.get_num_states(), regression_function.get_degree()),
            dtype=FLOAT, device=device)

    def _n_extra_coeffs(self) -> int:
        """doubles this product parks in the coefficient array besides regression coefficients"""
        return 0

    def supports_analytic_pv(self, model) -> bool:
        return False

    def supports_analytic_exposure(self, model) -> bool:
        return False

    def compute_pv_analytically(self, model):
        raise NotImplementedError

    # ---- native hooks --------------------------------------------------------------------------------------------
    def _cash_events(self, ctx) -> list[CashEvent]:
        """one CashEvent per entry of product_timeline, in order"""
        raise NotImplementedError

    def _value_terms(self, ctx, time: float) -> list:
        """linear combination of atoms giving this product's value observed at `time` (get_value in the reference)"""
        raise NotImplementedError

    def _observed_from(self, observation_date: float):
        """the underlying as re-struck at an observation date (generate_underlying_requests_for_date)"""
        raise NotImplementedError

    def generate_underlying_requests_for_date(self, observation_date: float) -> UnderlyingRequest:
        return UnderlyingRequest(self._observed_from(observation_date))
