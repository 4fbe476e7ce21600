"""Unilateral CVA = (1-R) * E[ sum_k relu(E_k) S(0,t_k) (1 - S(t_k,t_{k+1})) ]  (reference: metrics/cva_metric.py:7-100).
The per-path sum and its reduction run in csrc/k4_reduce.hip; this class registers the survival requests."""
from collections import defaultdict

from ..request_interface.request_types import AtomicRequest, AtomicRequestType
from .metric import Metric, MetricType


class CVAMetric(Metric):
    _native = True

    def __init__(self, counterparty_id: str, recovery_rate: float, evaluation_type=Metric.EvaluationType.NUMERICAL):
        super().__init__(MetricType.CVA, evaluation_type)
        self.counterparty_id = counterparty_id
        self.recovery_rate = recovery_rate
        self.survival_prob_requests: dict = {}
        self.cond_survival_prob_requests: dict = {}

    def get_counterparty_ids(self):
        return [self.counterparty_id]

    def get_name(self) -> str:
        return f"cva[{self.counterparty_id}]"

    def set_requests(self, exposure_timeline) -> None:
        for idx in range(len(exposure_timeline) - 1):
            label = (idx, self.counterparty_id)
            self.cond_survival_prob_requests[label] = AtomicRequest(
                AtomicRequestType.CONDITIONAL_SURVIVAL_PROBABILITY,
                time1=exposure_timeline[idx], time2=exposure_timeline[idx + 1])
            self.survival_prob_requests[label] = AtomicRequest(AtomicRequestType.SURVIVAL_PROBABILITY)

    def get_requests(self):
        out = defaultdict(list)
        for label, req in self.survival_prob_requests.items():
            out[label].append(req)
        for label, req in self.cond_survival_prob_requests.items():
            out[label].append(req)
        return out
