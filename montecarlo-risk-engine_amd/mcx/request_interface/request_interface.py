"""RequestInterface (reference surface: request_interface/request_interface.py:9-130).

The reference resolves every de-duplicated request into an [N] vector before valuation (~2.4 GB at 1 M paths).  Here the
built-in products/metrics never materialise them: a request is an *atom* (include/mcx.h) evaluated in registers by the
kernels.  This class keeps the collection / handle-assignment API and, for pluggable Metric subclasses, returns lazily
materialised vectors (`mcx_resolve_atoms`)."""
from __future__ import annotations

from collections import defaultdict
from numbers import Integral

import torch


class _LazyResolved:
    """sequence indexed by handle; materialises the vector on first access"""

    def __init__(self, resolver, n):
        self._resolver, self._cache, self._n = resolver, {}, n

    def __len__(self):
        return self._n

    def __getitem__(self, handle):
        if handle not in self._cache:
            self._cache[handle] = self._resolver(handle)
        return self._cache[handle]


class RequestInterface:
    def __init__(self, model):
        self.model = model
        self.num_atomic_requests = 0
        self.num_composite_requests = 0
        self.all_requests = defaultdict(set)
        self.all_composite_requests = defaultdict(set)
        self._handle_to_key = {}       # handle -> (time_index, asset_id, request)
        self._comp_handle_to_req = {}
        self._timeline = None

    def _register(self, req, table, key, counter, store):
        if key not in table:
            table[key] = counter
            store[counter] = key
            counter += 1
        req.set_handle(table[key])
        return counter

    def collect_and_index_requests(self, products, simulation_timeline, exposure_requests, exposure_timeline):
        self._timeline = [float(t) for t in simulation_timeline]
        time_to_index = {t: i for i, t in enumerate(self._timeline)}
        atomic, comp = {}, {}
        n_atomic = n_comp = 0
        for prod in products:
            for und_time, und_reqs in prod.get_underlying_requests().items():
                ti = time_to_index[float(prod.modeling_timeline[und_time])]
                for ur in und_reqs:
                    self.all_composite_requests[ti].add(ur)
                    n_comp = self._register(ur, comp, (ti, ur), n_comp, self._comp_handle_to_req)
                    for label, reqs in ur.get_atomic_requests().items():
                        for req in reqs:
                            self.all_requests[(ti, label[1])].add(req)
                            n_atomic = self._register(req, atomic, (ti, label[1], req), n_atomic, self._handle_to_key)
        for prod in products:
            for (t, asset_id), reqs in prod.get_atomic_requests().items():
                ti = time_to_index[float(prod.modeling_timeline[t])]
                for req in reqs:
                    self.all_requests[(ti, asset_id)].add(req)
                    n_atomic = self._register(req, atomic, (ti, asset_id, req), n_atomic, self._handle_to_key)
        for (t, asset_id), reqs in exposure_requests.items():
            time = float(exposure_timeline[t]) if isinstance(t, Integral) else float(t)
            ti = time_to_index[time]
            for req in reqs:
                self.all_requests[(ti, asset_id)].add(req)
                n_atomic = self._register(req, atomic, (ti, asset_id, req), n_atomic, self._handle_to_key)
        self.num_atomic_requests, self.num_composite_requests = n_atomic, n_comp

    def resolve_requests(self, paths: torch.Tensor):
        """paths: [N, T, D] view as returned by MonteCarloEngine.generate_paths (or native [T, D, N])"""
        def atomic(handle):
            ti, asset_id, req = self._handle_to_key[handle]
            state = paths[:, ti] if paths.shape[0] != len(self._timeline) or paths.ndim != 3 else paths[ti].T
            return self.model.resolve_request(req, asset_id, state)

        resolved = _LazyResolved(atomic, self.num_atomic_requests)

        def composite(handle):
            _, ur = self._comp_handle_to_req[handle]
            return ur.get_value(resolved)

        return [resolved, _LazyResolved(composite, self.num_composite_requests)]
