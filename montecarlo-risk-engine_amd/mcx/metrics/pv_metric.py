"""Present value of a netting set.

NUMERICAL evaluation: the mean over paths of the summed, numeraire-normalised cashflows together with its Monte-Carlo standard
error.  The per-path sums never reach Python — the fused kernel (csrc/kf_fused.hip) or `mcx_reduce_vector` accumulates
(n, shift, sum, sum of squares) records per GPU and `metric.mean_and_error` merges them across ranks.
ANALYTICAL evaluation: the product's closed form (`compute_pv_analytically`), no simulation; under `differentiate=True` the
closed form is differentiated with autograd (mcx/aad.py)."""
import torch

from .metric import Metric, MetricType


class PVMetric(Metric):
    _native = True          # reduced by the library's own kernels, not handed to a plugin

    def __init__(self, evaluation_type=Metric.EvaluationType.NUMERICAL):
        super().__init__(MetricType.PV, evaluation_type)

    def __repr__(self):
        return f"PVMetric({self.evaluation_type.name.lower()})"

    def evaluate_analytically(self, product=None, model=None, **kwargs):
        if model is None or product is None:
            raise ValueError("Analytical PV evaluation requires both product and model.")
        value = product.compute_pv_analytically(model).reshape(-1)[0]
        return [(value, torch.zeros((), dtype=value.dtype))]
