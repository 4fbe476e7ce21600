"""Expected negative exposure per date (reference: metrics/ene_metric.py:3-16)."""
from .metric import Metric, MetricType


class ENEMetric(Metric):
    _native = True

    def __init__(self, evaluation_type=Metric.EvaluationType.NUMERICAL):
        super().__init__(MetricType.ENE, evaluation_type)
