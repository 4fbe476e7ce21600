"""Metric base class (reference surface: metrics/metric.py:7-60).

Built-in metrics are evaluated by fused HIP reductions (csrc/k4_reduce.hip, k5_select.hip); this module keeps the
Metrics *API* (types, names, request registration, `evaluate` signature) and the host-side finalisation of the
accumulator records the kernels return (mean and Monte-Carlo error, metric.py:26-35)."""
from __future__ import annotations

import math
from collections import defaultdict
from enum import Enum

import numpy as np
import torch

from ..common.packages import FLOAT, device


class MetricType(Enum):
    PV = "Present Value"
    CE = "Current Exposure"
    EPE = "Expected Positive Exposure"
    ENE = "Expected Negative Exposure"
    PFE = "Potential Future Exposure"
    EEPE = "Effective Expected Positive Exposure"
    CVA = "Credit Valuation Adjustment"


def combine_acc(recs: np.ndarray) -> tuple[float, float, float]:
    """Merge per-shard accumulator records (n, shift, s1, s2) -> (N, mean, M2) with Chan's pairwise update, so shards
    (GPUs) may use different shifts and no catastrophic cancellation is introduced."""
    N, mean, M2 = 0.0, 0.0, 0.0
    for n, shift, s1, s2 in np.asarray(recs, dtype=np.float64).reshape(-1, 4):
        if n <= 0:
            continue
        m = shift + s1 / n
        q = s2 - s1 * s1 / n
        if q < 0.0:
            q = 0.0
        if N == 0:
            N, mean, M2 = n, m, q
        else:
            delta = m - mean
            tot = N + n
            mean = mean + delta * n / tot
            M2 = M2 + q + delta * delta * N * n / tot
            N = tot
    return N, mean, M2


def mean_and_error(recs: np.ndarray) -> tuple[float, float]:
    """metric.py:26-35: mean, std(unbiased)/sqrt(N)  (N == 1 gives NaN, as torch does)"""
    N, mean, M2 = combine_acc(recs)
    if N <= 1:
        return mean, float("nan")
    return mean, math.sqrt(M2 / (N - 1.0)) / math.sqrt(N)


class Metric:
    class EvaluationType(Enum):
        ANALYTICAL = "Analytical"
        NUMERICAL = "Numerical"

    def __init__(self, metric_type, evaluation_type):
        self.metric_type = metric_type
        self.evaluation_type = evaluation_type

    def _compute_mc_mean_and_error(self, values: torch.Tensor):
        """small-tensor host version kept for pluggable subclasses (metric.py:26-35)"""
        n = values.shape[0]
        return values.mean(), values.std(unbiased=True) / math.sqrt(n)

    def set_requests(self, exposure_timeline) -> None:
        pass

    def get_requests(self):
        return defaultdict(list)

    def get_counterparty_ids(self):
        return None

    def get_name(self) -> str:
        return self.metric_type.name.lower()

    def evaluate_analytically(self, **kwargs):
        raise NotImplementedError("Analytical evaluation not implemented.")

    def evaluate_numerically(self, **kwargs):
        raise NotImplementedError("Numerical evluation not implemented.")

    def evaluate(self, **kwargs):
        if self.evaluation_type == Metric.EvaluationType.NUMERICAL:
            return self.evaluate_numerically(**kwargs)
        return self.evaluate_analytically(**kwargs)

    # True for the library's own metrics, whose reductions run fused on the GPU; user subclasses that override
    # evaluate_numerically are handed torch views of the device buffers instead (SURVEY.md §8b "Metrics API").
    _native = False
