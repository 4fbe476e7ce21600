"""Metrics API as a plugin surface (reference: metrics/metric.py:37-60, controller.py:554-562): a user-defined Metric subclass
receives `exposures` (list of [N] tensors after netting/threshold/collateral), `cfs`, `resolved_requests` indexed by
request handle, the netting set and the model — as torch views of the backend's buffers."""
import numpy as np
import pytest
import torch

import cases
from mcx.controller.controller import SimulationController
from mcx.metrics.cva_metric import CVAMetric
from mcx.metrics.metric import Metric, MetricType
from mcx.metrics.risk_metrics import RiskMetrics


class MeanSquareExposure(Metric):
    def __init__(self):
        super().__init__(MetricType.EPE, Metric.EvaluationType.NUMERICAL)

    def get_name(self):
        return "mse"

    def evaluate_numerically(self, exposures, **kwargs):
        return [self._compute_mc_mean_and_error(e * e) for e in exposures]


class UserCVA(CVAMetric):
    """the CVA formula written by a user against the Metrics API (cva_metric.py:62-100), NOT the fused kernel"""
    _native = False

    def get_name(self):
        return "user_cva"

    def evaluate_numerically(self, exposures, resolved_requests, **kwargs):
        total = torch.zeros_like(exposures[0])
        for k in range(len(exposures) - 1):
            label = (k, self.counterparty_id)
            surv = resolved_requests[0][self.survival_prob_requests[label].handle]
            cond = resolved_requests[0][self.cond_survival_prob_requests[label].handle]
            total = total + torch.relu(exposures[k]) * surv * (1 - cond)
        return [self._compute_mc_mean_and_error(total * (1.0 - self.recovery_rate))]


def _run(backend):
    # (two CVA metrics on one counterparty cannot coexist: equal requests are de-duplicated and only one object receives
    #  its handle — in the reference too, controller.py:246-249 — so the native metric runs in a controller of its own)
    out = []
    for metrics in ([UserCVA("cp", 0.4), MeanSquareExposure()], [CVAMetric("cp", 0.4)]):
        ns, model, _ = cases.irs_cva()
        rm = RiskMetrics(metrics, exposure_timeline=np.arange(11) * 0.25)
        sc = SimulationController(ns, model, rm, 4096, 2048, 2, cases.E, backend=backend)
        out.append((sc, sc.run_simulation()))
    return out


def _check(runs):
    (sc, res), (sc_n, res_n) = runs
    assert sc._fused is None and sc_n._fused is not None     # a non-native metric forces the materialising plan
    user, native = res.results[0][0][0], res_n.results[0][0][0]
    assert np.isclose(native[0], user[0], rtol=1e-12) and np.isclose(native[1], user[1], rtol=1e-9)
    expo = sc.last_state["expo"][0].cpu().numpy()
    mse = np.array([v for v, _ in res.results[0][1]])
    assert np.allclose(mse, (expo ** 2).mean(axis=1), rtol=1e-12)
    assert res.metric_names == ["user_cva", "mse"]


def test_plugin_metric_host_logic(oracle):
    _check(_run(oracle))


@pytest.mark.gpu
def test_plugin_metric_gpu(hip):
    _check(_run(hip))
