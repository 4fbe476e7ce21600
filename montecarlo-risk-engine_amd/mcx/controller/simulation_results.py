"""SimulationResults: numpy container with named / legacy-keyword access
(reference surface: controller/simulation_results.py:5-338).

results[netting_set][metric][evaluation] = (value, mc_error); derivatives[...][evaluation][param];
second_derivatives[...][evaluation][param1][param2]. Names resolve case-insensitively."""
from __future__ import annotations

import numpy as np
import torch

_NS_ALIASES = ("prod_idx", "product", "product_idx")
_METRIC_ALIASES = ("metric_idx", "metric_set_idx")
_EVAL_ALIASES = ("evaluation_index",)


def _to_numpy(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu().numpy()
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_numpy(x) for x in obj)
    return obj


class SimulationResults:
    def __init__(self, results, derivatives, second_derivatives, netting_set_names=None, metric_names=None,
                 model_param_names=None, product_names=None):
        self.results = _to_numpy(results)
        self.derivatives = _to_numpy(derivatives)
        self.second_derivatives = _to_numpy(second_derivatives)
        n_ns = len(self.results)
        n_metrics = len(self.results[0]) if n_ns else 0
        if netting_set_names is not None and product_names is not None and netting_set_names != product_names:
            raise ValueError("Provide either 'netting_set_names' or legacy alias 'product_names', not conflicting values.")
        names = netting_set_names if netting_set_names is not None else product_names
        self.netting_set_names = names if names is not None else [f"netting_set_{i}" for i in range(n_ns)]
        self.product_names = self.netting_set_names
        self.metric_names = metric_names if metric_names is not None else [f"metric_{i}" for i in range(n_metrics)]
        self.model_param_names = model_param_names if model_param_names is not None else []
        self._ns_idx = {n.lower(): i for i, n in enumerate(self.netting_set_names)}
        self._metric_idx = {n.lower(): i for i, n in enumerate(self.metric_names)}
        self._param_idx = {n.lower(): i for i, n in enumerate(self.model_param_names)}

    # ---- argument handling ---------------------------------------------------------------------------------------
    @staticmethod
    def _pop_alias(kwargs: dict, aliases, label):
        value = None
        for a in aliases:
            if a in kwargs:
                v = kwargs.pop(a)
                if value is None:
                    value = v
                elif v != value:
                    raise ValueError(f"Conflicting values provided for '{label}' and legacy alias '{a}'.")
        return value

    def _locate(self, netting_set, metric, evaluation_idx, legacy: dict):
        legacy_ns = self._pop_alias(legacy, _NS_ALIASES, "netting_set")
        legacy_metric = self._pop_alias(legacy, _METRIC_ALIASES, "metric")
        legacy_eval = self._pop_alias(legacy, _EVAL_ALIASES, "evaluation_idx")
        if legacy:
            raise TypeError("Unexpected keyword argument(s): " + ", ".join(sorted(legacy)))
        netting_set = legacy_ns if netting_set is None else netting_set
        metric = legacy_metric if metric is None else metric
        evaluation_idx = legacy_eval if evaluation_idx is None else evaluation_idx
        return self._lookup(netting_set, self._ns_idx, "netting set", self.netting_set_names), \
            self._lookup(metric, self._metric_idx, "metric", self.metric_names), evaluation_idx

    @staticmethod
    def _lookup(key, table, what, available):
        if isinstance(key, str):
            if key.lower() not in table:
                raise KeyError(f"Unknown {what} name '{key}'. Available: {available}")
            return table[key.lower()]
        return key

    def _param(self, p):
        return self._lookup(p, self._param_idx, "model parameter", self.model_param_names)

    # ---- accessors -----------------------------------------------------------------------------------------------
    def get_product_names(self):
        return list(self.netting_set_names)

    def get_netting_set_names(self):
        return list(self.netting_set_names)

    def get_metric_names(self):
        return list(self.metric_names)

    def get_model_param_names(self):
        return list(self.model_param_names)

    def _column(self, netting_set, metric, evaluation_idx, legacy, col):
        ns, m, ev = self._locate(netting_set, metric, evaluation_idx, legacy)
        vals = np.array([r[col] for r in self.results[ns][m]])
        return vals if ev is None else vals[ev]

    def get_results(self, netting_set=None, metric=None, evaluation_idx=None, **legacy_kwargs):
        return self._column(netting_set, metric, evaluation_idx, legacy_kwargs, 0)

    def get_mc_error(self, netting_set=None, metric=None, evaluation_idx=None, **legacy_kwargs):
        return self._column(netting_set, metric, evaluation_idx, legacy_kwargs, 1)

    def get_derivatives(self, netting_set=None, metric=None, param=None, evaluation_idx=None, **legacy_kwargs):
        ns, m, ev = self._locate(netting_set, metric, evaluation_idx, legacy_kwargs)
        d = self.derivatives[ns][m]
        if param is None and ev is None:
            return d
        if ev is not None:
            d = d[ev]
            if param is None:
                return {name: d[i] for i, name in enumerate(self.model_param_names)}
            return d[self._param(param)]
        pi = self._param(param)
        return np.array([e[pi] for e in d])

    def get_second_derivatives(self, netting_set=None, metric=None, param1=None, param2=None, evaluation_idx=None,
                               **legacy_kwargs):
        ns, m, ev = self._locate(netting_set, metric, evaluation_idx, legacy_kwargs)
        h = self.second_derivatives[ns][m]
        if param1 is None and param2 is None and ev is None:
            return h
        names = self.model_param_names

        def row(r):
            return {n: r[i] for i, n in enumerate(names)}

        if ev is not None:
            h = h[ev]
            if param1 is None and param2 is None:
                return {n: row(h[i]) for i, n in enumerate(names)}
            if param2 is None:
                return row(h[self._param(param1)])
            if param1 is None:
                c = self._param(param2)
                return {n: h[i][c] for i, n in enumerate(names)}
            return h[self._param(param1)][self._param(param2)]
        if param1 is not None and param2 is not None:
            r, c = self._param(param1), self._param(param2)
            return np.array([e[r][c] for e in h])
        raise ValueError("When evaluation_idx is omitted, provide both param1 and param2 or neither.")
