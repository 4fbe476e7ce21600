"""Collection of metrics + the pathwise primitives they need (reference: metrics/risk_metrics.py:9-69)."""
from __future__ import annotations

from enum import Enum

import numpy as np
import torch

from ..common.packages import FLOAT, device
from .metric import Metric, MetricType


class PathwisePrimitive(Enum):
    DISCOUNTED_CASHFLOWS = "discounted_cashflows"
    EXPOSURE_PROFILES = "exposure_profiles"


class RiskMetrics:
    def __init__(self, metrics: list[Metric], exposure_timeline=None):
        self.metrics = metrics
        if exposure_timeline is None:
            exposure_timeline = []
        self.exposure_timeline = torch.tensor(np.asarray(exposure_timeline, dtype=np.float64), dtype=FLOAT, device=device)
        self.any_pv = any(m.metric_type == MetricType.PV for m in metrics)
        self.any_xva = any(m.metric_type == MetricType.CVA for m in metrics)
        self.any_exposure = any(m.metric_type != MetricType.PV for m in metrics)
        prims = []
        if self.any_pv:
            prims.append(PathwisePrimitive.DISCOUNTED_CASHFLOWS)
        if self.any_exposure:
            prims.append(PathwisePrimitive.EXPOSURE_PROFILES)
        self._required_primitives = frozenset(prims)
        if self.any_exposure:
            assert len(exposure_timeline) > 0, \
                "For exposure simulation at least one exposure time point needs to be provided."
        for m in self.metrics:
            m.set_requests(exposure_timeline)
        self.counterparty_ids: list[str] = []
        for m in self.metrics:
            ids = m.get_counterparty_ids()
            if ids is not None:
                self.counterparty_ids.extend(ids)

    def requires_discounted_cashflows(self) -> bool:
        return self.requires_primitive(PathwisePrimitive.DISCOUNTED_CASHFLOWS)

    def requires_exposure_profiles(self) -> bool:
        return self.requires_primitive(PathwisePrimitive.EXPOSURE_PROFILES)

    def required_pathwise_primitives(self):
        return self._required_primitives

    def requires_primitive(self, primitive: PathwisePrimitive) -> bool:
        return primitive in self._required_primitives
