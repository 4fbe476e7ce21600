"""Netting set: products valued on a netted basis, with symmetric threshold and margin-period-of-risk collateral
(reference: products/netting_set.py:12-184). The per-path transforms run in the reduction kernels' prologue; this class
only carries the description (`_unsecured_spec`)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence

import torch

from .product import Product


@dataclass
class NettingSet:
    name: str
    products: Sequence[Product]
    threshold: float = 0.0
    margin_period_of_risk: float | None = None
    counterparty_id: str | None = None
    collateral_interpolation: str = "linear"

    def __post_init__(self):
        self.products = list(self.products)
        if len(self.products) == 0:
            raise ValueError("A netting set must contain at least one product.")
        if self.threshold < 0.0:
            raise ValueError("Netting set threshold must be non-negative.")
        if self.margin_period_of_risk is not None and self.margin_period_of_risk < 0.0:
            raise ValueError("Netting set margin period of risk must be non-negative.")
        if self.collateral_interpolation not in {"linear", "previous"}:
            raise ValueError("Collateral interpolation must be one of {'linear', 'previous'}.")

    def get_name(self) -> str:
        return self.name

    def is_collateralized(self) -> bool:
        return self.margin_period_of_risk is not None

    def get_collateral_query_times(self, exposure_timeline: torch.Tensor) -> torch.Tensor:
        if not self.is_collateralized():
            return torch.zeros(0, dtype=exposure_timeline.dtype)
        delayed = exposure_timeline - self.margin_period_of_risk
        return delayed[delayed >= 0.0]

    # host-side restatements on small tensors (API parity; netting_set.py:48-72, 110-184) ------------------------
    def apply_threshold(self, exposures: torch.Tensor) -> torch.Tensor:
        if exposures.numel() == 0 or self.threshold == 0.0:
            return exposures
        h = self.threshold
        return torch.where(exposures > h, exposures - h,
                           torch.where(exposures < -h, exposures + h, torch.zeros_like(exposures)))

    def _exposure_at(self, netted_exposures: torch.Tensor, exposure_timeline: torch.Tensor, query_times: torch.Tensor):
        """netted exposure rows looked up at arbitrary times (netting_set.py:76-107): the last grid date at or before the query
        ('previous') or the straight line between the two neighbouring grid dates ('linear'); zero before the first grid date,
        flat after the last one"""
        if netted_exposures.numel() == 0:
            return netted_exposures
        last = exposure_timeline.shape[0] - 1
        early = (query_times < exposure_timeline[0]).unsqueeze(1)
        if self.collateral_interpolation == "previous":
            at = (torch.searchsorted(exposure_timeline, query_times, right=True) - 1).clamp(0, last)
            vals = netted_exposures.index_select(0, at)
        else:
            hi = torch.searchsorted(exposure_timeline, query_times).clamp(max=last)
            lo = (hi - 1).clamp(min=0)
            t_lo, t_hi = exposure_timeline.index_select(0, lo), exposure_timeline.index_select(0, hi)
            span = t_hi - t_lo
            w = torch.where(span > 0.0, (query_times - t_lo) / span, torch.zeros_like(query_times)).unsqueeze(1)
            v_lo = netted_exposures.index_select(0, lo)
            vals = v_lo + w * (netted_exposures.index_select(0, hi) - v_lo)
        return torch.where(early, torch.zeros_like(vals), vals)

    def compute_collateral_profile(self, netted_exposures, exposure_timeline, metric_exposure_indices=None,
                                   delayed_exposure_indices=None):
        if metric_exposure_indices is None or delayed_exposure_indices is None:
            # no exact delayed grid dates: the collateral call is the thresholded exposure interpolated at t - MPoR
            # (netting_set.py:151-157).  SimulationController never takes this branch (its internal grid contains every
            # delayed date, controller.py:153-162); it is API parity for direct callers of the Metrics API
            if metric_exposure_indices is not None:
                rows = netted_exposures.index_select(0, metric_exposure_indices)
                if not self.is_collateralized() or netted_exposures.numel() == 0:
                    return torch.zeros_like(rows)
            if not self.is_collateralized() or netted_exposures.numel() == 0:
                return torch.zeros_like(netted_exposures)
            delayed = self._exposure_at(netted_exposures, exposure_timeline, exposure_timeline - self.margin_period_of_risk)
            return self.apply_threshold(delayed)
        rows = netted_exposures.index_select(0, metric_exposure_indices)
        out = torch.zeros_like(rows)
        if not self.is_collateralized() or netted_exposures.numel() == 0:
            return out
        valid = delayed_exposure_indices >= 0
        if torch.any(valid):
            out[valid] = self.apply_threshold(netted_exposures.index_select(0, delayed_exposure_indices[valid]))
        return out

    def compute_unsecured_exposure_profiles(self, netted_exposures, exposure_timeline, metric_exposure_indices=None,
                                            delayed_exposure_indices=None):
        if netted_exposures.numel() == 0:
            return netted_exposures
        rows = (netted_exposures.index_select(0, metric_exposure_indices)
                if metric_exposure_indices is not None else netted_exposures)
        if not self.is_collateralized():
            return self.apply_threshold(rows)
        return rows - self.compute_collateral_profile(netted_exposures, exposure_timeline, metric_exposure_indices,
                                                      delayed_exposure_indices)
