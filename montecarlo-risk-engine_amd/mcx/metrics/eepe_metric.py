"""'Effective' EPE exactly as the reference defines it: the plain time-average of the EE profile with the
std-over-dates / sqrt(#dates) as error — no running maximum (reference: metrics/eepe_metric.py:3-16)."""
from .metric import Metric, MetricType


class EEPEMetric(Metric):
    _native = True

    def __init__(self, evaluation_type=Metric.EvaluationType.NUMERICAL):
        super().__init__(MetricType.EEPE, evaluation_type)
