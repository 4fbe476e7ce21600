"""Current exposure: positive part of the first exposure date (reference: metrics/ce_metric.py:3-13)."""
from .metric import Metric, MetricType


class CEMetric(Metric):
    _native = True

    def __init__(self, evaluation_type=Metric.EvaluationType.NUMERICAL):
        super().__init__(MetricType.CE, evaluation_type)
