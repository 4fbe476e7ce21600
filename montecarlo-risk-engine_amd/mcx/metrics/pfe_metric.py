"""Potential future exposure: exact order statistic x_(ceil(qN)-1) of the signed netted exposure with the reference's
finite-difference density error (reference: metrics/pfe_metric.py:4-73). On the GPU the order statistics come from a
radix select (no sort)."""
import math

import numpy as np

from .metric import Metric, MetricType


class PFEMetric(Metric):
    _native = True

    def __init__(self, quantile=0.95, evaluation_type=Metric.EvaluationType.NUMERICAL):
        super().__init__(MetricType.PFE, evaluation_type)
        self.quantile = quantile

    def get_name(self) -> str:
        return f"pfe[{self.quantile:g}]"

    def q_index(self, num_paths: int) -> int:
        # the reference computes ceil() on a float32 tensor (torch.tensor(python_float) is float32): pfe_metric.py:59
        return int(np.ceil(np.float32(self.quantile * num_paths))) - 1

    def quantile_error(self, lo: float, mid: float, hi: float, q_index: int, num_paths: int) -> float:
        """pfe_metric.py:13-44"""
        if q_index == 0 or q_index == num_paths - 1:
            return 0.0
        if lo == mid and hi == mid:
            return 0.0
        f_q = max((hi - lo) / 2.0, 1e-6)
        return math.sqrt(self.quantile * (1 - self.quantile) / (num_paths * f_q * f_q))
