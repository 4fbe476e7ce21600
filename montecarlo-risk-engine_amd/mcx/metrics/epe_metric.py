"""Expected positive exposure per date (reference: metrics/epe_metric.py:3-16)."""
from .metric import Metric, MetricType


class EPEMetric(Metric):
    _native = True

    def __init__(self, evaluation_type=Metric.EvaluationType.NUMERICAL):
        super().__init__(MetricType.EPE, evaluation_type)
