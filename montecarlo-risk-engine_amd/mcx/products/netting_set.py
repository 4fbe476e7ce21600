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

    def compute_collateral_profile(self, netted_exposures, exposure_timeline, metric_exposure_indices=None,
                                   delayed_exposure_indices=None):
        if metric_exposure_indices is None or delayed_exposure_indices is None:
            raise NotImplementedError("collateral profiles are evaluated on exact delayed exposure indices")
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
