"""Single / double barrier option with discrete monitoring and fuzzy barrier indicators (reference:
products/barrier_option.py:15-125, 298-314).  GPU: one MCX_EV_OPTION event in aggregation mode 4: its terms are the spot at
every observation date (running max / min in registers), `x_atom` the spot at maturity (include/mcx.h).  The payoff is
normalised by the numeraire of the FIRST observation date exactly like the reference (:310).

`set_use_brownian_bridge()` (:126-223) adds the bridge crossing correction (mode 5): one uniform per monitored interval and
barrier, drawn inside the kernels from the Philox stream (the reference draws them from a numpy Generator on the host; a
parity run injects those numbers, mcx_book_set_bridge_rng)."""
from __future__ import annotations

from enum import Enum
from typing import Optional

import torch

from .. import _abi
from ..common.packages import FLOAT, device
from ..request_interface.request_types import AtomicRequest, AtomicRequestType
from .product import CashEvent, OptionType, Product, ProductFamily


class BarrierOptionType(Enum):
    DOWNANDOUT = "Down-And-Out"
    UPANDOUT = "Up-And-Out"
    DOWNANDIN = "Down-And-In"
    UPANDIN = "Up-And-In"


_CODE = {BarrierOptionType.UPANDOUT: 1, BarrierOptionType.DOWNANDOUT: 2, BarrierOptionType.UPANDIN: 3, BarrierOptionType.DOWNANDIN: 4}


class BarrierOption(Product):
    def __init__(self, startdate: float, maturity: float, strike: float, num_observation_timepoints: int,
                 option_type: OptionType, barrier1: float, barrier_option_type1: Optional[BarrierOptionType],
                 barrier2: Optional[float] = None, barrier_option_type2: Optional[BarrierOptionType] = None,
                 asset_id: str | None = None):
        super().__init__(asset_ids=[asset_id], product_family=ProductFamily.BARRIER_PATH_TERMINAL)
        self._T, self._K = float(maturity), float(strike)
        self.strike = torch.tensor([self._K], dtype=FLOAT, device=device)
        self.maturity = torch.tensor([self._T], dtype=FLOAT, device=device)
        self.product_timeline = torch.tensor([self._T], dtype=FLOAT, device=device)
        self.modeling_timeline = torch.linspace(startdate, maturity, num_observation_timepoints, dtype=FLOAT, device=device)
        self.regression_timeline = torch.tensor([], dtype=FLOAT, device=device)
        self.barrier1 = torch.tensor([float(barrier1)], dtype=FLOAT, device=device)
        self.barrier_option_type1 = barrier_option_type1
        self.barrier2 = None if barrier2 is None else torch.tensor([float(barrier2)], dtype=FLOAT, device=device)
        self.barrier_option_type2 = barrier_option_type2
        self.option_type = option_type
        self.use_brownian_bridge = False
        self.numeraire_requests = {i: AtomicRequest(AtomicRequestType.NUMERAIRE, float(t))
                                   for i, t in enumerate(self.modeling_timeline)}
        self.spot_requests = {(i, self.get_asset_id()): AtomicRequest(AtomicRequestType.SPOT)
                              for i in range(len(self.modeling_timeline))}

    def set_use_brownian_bridge(self):
        self.use_brownian_bridge = True

    def _n_extra_coeffs(self) -> int:
        return 2 if self.use_brownian_bridge else 0

    def _cash_events(self, ctx):
        if self.barrier_option_type1 not in _CODE:
            raise NotImplementedError(f"Barrier type {self.barrier_option_type1} not supported.")
        obs = [float(t) for t in self.modeling_timeline]
        terms = [(1.0, ctx.atom(AtomicRequest(AtomicRequestType.SPOT), self.get_asset_id(), t)) for t in obs]
        two = self.barrier2 is not None and self.barrier_option_type2 is not None
        types = _CODE[self.barrier_option_type1] + (8 * _CODE[self.barrier_option_type2] if two else 0)
        aux = (5.0 if self.use_brownian_bridge else 4.0, float(self.barrier1[0]), float(self.barrier2[0]) if two else 0.0, float(types))
        params = ()
        if self.use_brownian_bridge:
            from ..models.black_scholes import BlackScholesModel
            if not isinstance(ctx.model, BlackScholesModel):
                raise NotImplementedError("the Brownian-bridge correction reads model.get_volatility() of a single-asset Black-Scholes model")
            sigma = ctx.model._pf(1)
            # crossing probability exp(-2 ln(S_k/B) ln(S_k+1/B) / (sigma^2 * maturity / n_observations))   (barrier_option.py:146)
            params = (-2.0 / (sigma * sigma * (self._T / len(obs))), float(ctx.current_product))
        sign = 1.0 if self.option_type == OptionType.CALL else -1.0
        # the maturity spot is the LAST monitored value (paths[:, -1], :70): read it at the last observation date
        return [CashEvent(_abi.EV_OPTION, self._T, terms, strike=self._K, sign=sign, aux=aux, x_asset=self.get_asset_id(),
                          x_time=obs[-1], num_time=obs[len(self.product_timeline) - 1], coeff_params=params)]
