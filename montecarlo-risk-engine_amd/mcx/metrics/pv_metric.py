"""Present value (reference: metrics/pv_metric.py:3-18)."""
import torch

from .metric import Metric, MetricType


class PVMetric(Metric):
    _native = True

    def __init__(self, evaluation_type=Metric.EvaluationType.NUMERICAL):
        super().__init__(MetricType.PV, evaluation_type)

    def evaluate_analytically(self, product=None, model=None, **kwargs):
        if product is None or model is None:
            raise ValueError("Analytical PV evaluation requires both product and model.")
        pv = product.compute_pv_analytically(model).squeeze()
        return [(pv, torch.zeros_like(pv))]
