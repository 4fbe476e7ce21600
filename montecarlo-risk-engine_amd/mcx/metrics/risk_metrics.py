"""The set of metrics of a run and what they need from the simulation: discounted cashflows (PV) and / or exposure profiles
(everything else).  Same constructor and query methods as the reference's metrics/risk_metrics.py."""
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
        dates = np.asarray([] if exposure_timeline is None else exposure_timeline, dtype=np.float64).reshape(-1)
        kinds = {m.metric_type for m in metrics}
        self.metrics = metrics
        self.any_pv = MetricType.PV in kinds
        self.any_xva = MetricType.CVA in kinds
        self.any_exposure = bool(kinds - {MetricType.PV})
        if self.any_exposure and dates.size == 0:
            raise AssertionError("For exposure simulation at least one exposure time point needs to be provided.")
        self.exposure_timeline = torch.tensor(dates, dtype=FLOAT, device=device)
        self._required_primitives = frozenset(
            prim for prim, wanted in ((PathwisePrimitive.DISCOUNTED_CASHFLOWS, self.any_pv),
                                      (PathwisePrimitive.EXPOSURE_PROFILES, self.any_exposure)) if wanted)
        self.counterparty_ids: list[str] = []
        for metric in metrics:
            metric.set_requests([] if exposure_timeline is None else exposure_timeline)
            self.counterparty_ids += list(metric.get_counterparty_ids() or [])

    def required_pathwise_primitives(self):
        return self._required_primitives

    def requires_primitive(self, primitive: PathwisePrimitive) -> bool:
        return primitive in self._required_primitives

    def requires_discounted_cashflows(self) -> bool:
        return PathwisePrimitive.DISCOUNTED_CASHFLOWS in self._required_primitives

    def requires_exposure_profiles(self) -> bool:
        return PathwisePrimitive.EXPOSURE_PROFILES in self._required_primitives
