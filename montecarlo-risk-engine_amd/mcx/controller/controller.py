"""SimulationController — host orchestration of the MI355X Monte-Carlo path
(reference surface: controller/controller.py:21-709; same constructor, same run_simulation() -> SimulationResults).

What stays on the host (µs–ms): timeline unions keyed on exact floats, request/atom planning, the K x K regression
solves, metric finalisation.  What runs as hand-written HIP behind the C ABI (include/mcx.h):
  K1 path generation, K2 book evaluation (resolve + cashflows + exposures), K3 LSM normal equations,
  K4 EE/ENE/PV/CVA reductions, K5 PFE radix select.
Paths are sharded over the ranks of torch.distributed (one process per GPU); only accumulator records, normal equations
and select histograms cross GPUs (mcx/parallel.py)."""
from __future__ import annotations

import logging
import math
import threading
import time
from collections import defaultdict
from typing import Sequence

import numpy as np
import torch

from .. import _abi, _native
from ..common.enums import SimulationScheme
from ..common.packages import FLOAT, device
from ..engine.engine import MonteCarloEngine
from ..helpers.host_threads import single_threaded_host
from ..maths.regression import PolyomialRegression, RegressionFunction
from ..metrics.metric import Metric, MetricType, mean_and_error
from ..metrics.risk_metrics import PathwisePrimitive, RiskMetrics
from ..models.model import Model
from ..models.model_config import ModelConfig
from ..parallel import Shard
from ..plan import (BookCompiler, BookPlan, FusedPlan, SimPlan, UnsecuredSpec, solve_normal_equations,
                    solve_normal_equations_batch)
from ..products.netting_set import NettingSet
from ..products.product import Product
from ..request_interface.request_interface import RequestInterface
from ..request_interface.request_types import AtomicRequest, AtomicRequestType
from .simulation_results import SimulationResults

logger = logging.getLogger(__name__)

_SELECT_DIGITS = ((53, 11), (42, 11), (31, 11), (20, 11), (9, 11), (0, 9))


def _key_to_double(keys: np.ndarray) -> np.ndarray:
    keys = np.asarray(keys, dtype=np.uint64)
    top = np.uint64(1) << np.uint64(63)
    bits = np.where(keys & top, keys ^ top, ~keys)
    return bits.view(np.float64)


class SimulationController:
    @single_threaded_host()
    def __init__(self, netting_sets: Sequence[NettingSet], model: Model, risk_metrics: RiskMetrics,
                 num_paths_mainsim: int, num_paths_presim: int, num_steps: int, simulation_scheme: SimulationScheme,
                 differentiate: bool = False, regression_function: RegressionFunction = PolyomialRegression(degree=2),
                 *, backend=None, use_mfma: bool = False):
        self.risk_metrics = risk_metrics
        netting_sets = list(netting_sets)
        if len(netting_sets) == 0:
            raise ValueError("Provide at least one netting set.")
        seen: set[int] = set()
        for ns in netting_sets:
            for p in ns.products:
                if id(p) in seen:
                    raise ValueError("A product instance cannot belong to more than one netting set.")
                seen.add(id(p))
        products = [p for ns in netting_sets for p in ns.products]
        self.netting_sets = netting_sets
        self.product_to_netting_set_idx = [i for i, ns in enumerate(netting_sets) for _ in ns.products]

        self.metric_exposure_timeline = risk_metrics.exposure_timeline.clone()
        self.exposure_timeline = self._build_internal_exposure_timeline()
        self._exposure_time_to_idx = {float(t): i for i, t in enumerate(self.exposure_timeline)}
        self.metric_exposure_indices = torch.tensor(
            [self._exposure_time_to_idx[float(t)] for t in self.metric_exposure_timeline], dtype=torch.long)
        self.netting_set_delayed_exposure_indices = self._build_netting_set_delayed_exposure_indices()

        self.numeraire_requests = {(float(t), "numeraire"): AtomicRequest(AtomicRequestType.NUMERAIRE, time1=float(t))
                                   for t in self.exposure_timeline}
        self.spot_requests = {(float(t), a): AtomicRequest(AtomicRequestType.SPOT)
                              for p in products for a in p.asset_ids for t in self.exposure_timeline}
        if risk_metrics.any_xva:
            if not isinstance(model, ModelConfig):
                raise Exception("ModelConfig needs to be provided for xVA valuation.")
            if not all(cp in model.id_to_model for cp in risk_metrics.counterparty_ids):
                raise Exception("Not all models set for xVA valuation.")

        self.products = products
        self.model = model
        self.num_paths_presim = int(num_paths_presim)
        self.num_paths_mainsim = int(num_paths_mainsim)
        self.num_steps = int(num_steps)
        self.simulation_scheme = simulation_scheme
        self.differentiate = differentiate
        self.regression_function = regression_function
        self.requires_higher_order_derivatives = False
        self.use_mfma = use_mfma
        self.seed_offset = 0         # added to the reference's Philox keys 42 / 43: independent replications of a run
        # the reference accumulates LSM cashflows in a float32 cache (controller.py:312-330); reproduce it for parity
        self.reference_float32_cf_cache = True
        self.allow_fused = True      # fused event/metric program (csrc/kf_fused.hip) when the book is fusable
        # main-pass execution plan when fusable: "semi" = K1 writes the paths tensor, ONE kernel evaluates book + metrics
        # from it; "fused" = a single launch, nothing materialised; "auto" = fused when every date of the book compiled to a
        # straight-line record (the lean kernel), semi otherwise; (None/unfusable: K1, K2, K4 as separate launches)
        self.main_plan = "auto"
        self.materialize = False     # also write paths / cashflows / exposures in the fused pass (inspection, tests)
        self.batch_lsm = True        # product-batched LSM pre-simulation (one launch per backward step of the whole book)
        self.forward_mode = True     # differentiate=True: dual-number pass where it exists, bump-and-revalue otherwise
        # The reference compiles nothing: every run_simulation() sees the current state of its products / metrics.  Re-using the
        # compiled descriptors and the uploaded book of the previous run is an opt-in for callers that re-run an UNCHANGED
        # controller (bench.py, tools/run_configs.py); invalidate() drops the cache after a mutation.
        self.reuse_compiled = False
        # value polynomials (mcx_vpoly.hip): on by default, exact term loops when off
        self.collapse_values = True
        self.collapse_pad, self.collapse_rel_tol, self.collapse_min_terms = 0.3, 1e-14, 6
        self.n_collapsed_events = 0
        # PFE order statistics: one bracket pass over the exposure matrix + digit passes on the gathered candidates (csrc/k5_select.hip)
        self.bracket_select, self.bracket_sample = True, 16384
        self._backend = backend
        for i, p in enumerate(products):
            p.product_id = i
        if differentiate:
            self.model.requires_grad()

        K = regression_function.get_degree()
        if K > _abi.MAX_BASIS:
            raise ValueError(f"regression basis size {K} exceeds MCX_MAX_BASIS={_abi.MAX_BASIS}")
        self.regression_coeffs = []
        for p in products:
            p._allocate_regression_coeffs(regression_function)
            if p.get_num_states() > _abi.MAX_STATES:
                raise ValueError("too many exercise states")
            self.regression_coeffs.append(torch.zeros((len(self.exposure_timeline), p.get_num_states(), K), dtype=FLOAT))

        prod_times = {float(t) for p in products for t in p.modeling_timeline}
        exposure_times = {float(t) for t in self.exposure_timeline}
        self.simulation_timeline = torch.tensor(sorted(prod_times | exposure_times), dtype=FLOAT, device=device)
        self.requires_regression = any(self._product_requires_regression(p) for p in products)
        self.timings: dict[str, float] = {}
        self.last_state: dict = {}       # device buffers of the last run (paths / cfs / exposures) for inspection
        self._inject = {}                # tests: {"pre": (z, u), "main": (z, u)} recorded reference draws

    # ---- planning (host) -----------------------------------------------------------------------------------------
    @property
    def backend(self):
        if self._backend is None:
            self._backend = _native.get_backend()
        return self._backend

    def _build_internal_exposure_timeline(self) -> torch.Tensor:
        rm = self.risk_metrics
        if not rm.requires_exposure_profiles():
            return rm.exposure_timeline.clone()
        times = {float(t) for t in rm.exposure_timeline}
        for ns in self.netting_sets:
            if ns.is_collateralized():
                times.update(float(t) for t in ns.get_collateral_query_times(rm.exposure_timeline))
        return torch.tensor(sorted(times), dtype=FLOAT, device=device)

    def _build_netting_set_delayed_exposure_indices(self) -> list[torch.Tensor]:
        out = []
        for ns in self.netting_sets:
            idx = torch.full((len(self.metric_exposure_timeline),), -1, dtype=torch.long)
            if ns.is_collateralized():
                delayed = self.metric_exposure_timeline - ns.margin_period_of_risk
                for m, t in enumerate(delayed):
                    if float(t) >= 0.0:
                        idx[m] = self._exposure_time_to_idx[float(t)]
            out.append(idx)
        return out

    @staticmethod
    def _make_unique_names(base_names: list[str]) -> list[str]:
        counts: dict[str, int] = defaultdict(int)
        out = []
        for b in base_names:
            counts[b] += 1
            out.append(b if counts[b] == 1 else f"{b}#{counts[b]}")
        return out

    # The three per-product predicates are asked several times per product and run (compile, atom registration, regression, metric
    # evaluation): answers are kept per product object until the next compilation (`_compile` / `invalidate` clear the memo).  The
    # hit path is one dictionary lookup (no closure is built): books of 10^4-10^5 products ask ~6 times per product.
    def _can_use_analytic_exposure_for_product(self, product: Product) -> bool:
        memo = self.__dict__.setdefault("_pred_memo", {})
        key = ("analytic", id(product))
        hit = memo.get(key)
        if hit is None:
            ok = {MetricType.PV, MetricType.EPE, MetricType.PFE}
            hit = memo[key] = bool(all(m.metric_type in ok for m in self.risk_metrics.metrics) and product.supports_analytic_exposure(self.model))
        return hit

    def _product_requires_regression(self, product: Product) -> bool:
        memo = self.__dict__.setdefault("_pred_memo", {})
        key = ("regression", id(product))
        hit = memo.get(key)
        if hit is None:
            if len(self._timelines(product)[1]) > 0:
                hit = True
            elif not self.risk_metrics.requires_exposure_profiles():
                hit = False
            else:
                hit = not self._can_use_analytic_exposure_for_product(product)
            memo[key] = hit
        return hit

    def _can_skip_monte_carlo_for_product(self, product: Product) -> bool:
        memo = self.__dict__.setdefault("_pred_memo", {})
        key = ("skip", id(product))
        hit = memo.get(key)
        if hit is None:
            hit = memo[key] = bool((not self.risk_metrics.requires_exposure_profiles()) and all(
                m.metric_type == MetricType.PV and m.evaluation_type == Metric.EvaluationType.ANALYTICAL
                and product.supports_analytic_pv(self.model) for m in self.risk_metrics.metrics))
        return hit

    @staticmethod
    def _timelines(product: Product):
        """(payment dates, regression dates) of a product as tuples of floats, converted once per timeline tensor (a tensor ->
        list conversion per question was 5 conversions per product and run)"""
        hit = product.__dict__.get("_tl_tuples")
        pt, rt = product.product_timeline, product.regression_timeline
        if hit is None or hit[0] is not pt or hit[1] is not rt or hit[2] != (pt._version, rt._version):
            hit = product.__dict__["_tl_tuples"] = (pt, rt, (pt._version, rt._version), tuple(pt.tolist()), tuple(rt.tolist()))
        return hit[3], hit[4]

    def _get_requests(self):
        reqs = defaultdict(set)
        for label, r in self.numeraire_requests.items():
            reqs[label].add(r)
        for label, r in self.spot_requests.items():
            reqs[label].add(r)
        for m in self.risk_metrics.metrics:
            for label, rs in m.get_requests().items():
                reqs[label].update(rs)
        return reqs

    def compute_higher_derivatives(self):
        self.requires_higher_order_derivatives = True

    def _compile(self):
        """objects -> descriptors (BookPlan) + bookkeeping of coefficient offsets / metric atoms"""
        rm = self.risk_metrics
        sim_tl = [float(t) for t in self.simulation_timeline]
        K = self.regression_function.get_degree()
        comp = BookCompiler(self.model, sim_tl, K)
        self._pred_memo = {}
        self._sched_atom_cache = {}          # atom ids belong to ONE compiler: a re-compilation must not reuse the previous ones
        E = len(self.exposure_timeline)
        expo_times = [float(t) for t in self.exposure_timeline]
        want_expo = rm.requires_exposure_profiles()
        want_cfs = rm.requires_discounted_cashflows()
        prods = np.zeros(len(self.products), dtype=_abi.PRODUCT_DTYPE)
        self._expo_coeff_base, self._reg_coeff_base, self._cash_meta = [], [], []
        self._extra_coeff_base = []
        self._mc_products = []
        self._mc_set = set()
        off = 0
        for p_i, p in enumerate(self.products):
            S = p.get_num_states()
            self._expo_coeff_base.append(off)
            off += E * S * K
            self._reg_coeff_base.append(off)
            off += int(p.regression_timeline.shape[0]) * S * K          # (.shape: no tensor -> list conversion for 10^5 products)
            self._extra_coeff_base.append(off)
            off += p._n_extra_coeffs()
        expo_atom_cache: dict = {}

        def expo_atoms(asset):       # (numeraire, SPOT) atoms of every exposure date, once per asset
            hit = expo_atom_cache.get(asset)
            if hit is None:
                hit = [(comp.atom(self.numeraire_requests[(t, "numeraire")], "numeraire", t),
                        comp.atom(self.spot_requests[(t, asset)], asset, t)) for t in expo_times]
                expo_atom_cache[asset] = hit
            return hit

        expo_arr = np.asarray(expo_times, dtype=np.float64)
        num_atom_cache: dict = {}            # numeraire atom of a date: one lookup per distinct date instead of a request object per event
        expo_tmpl_cache: dict = {}
        W_EV = _abi.EVENT_DTYPE.itemsize // 8

        def expo_template(asset):    # EVENT_DTYPE rows of the exposure events of one asset, product-independent fields filled
            hit = expo_tmpl_cache.get(asset)
            if hit is None:
                hit = np.zeros(E, dtype=_abi.EVENT_DTYPE)
                at = expo_atoms(asset)
                hit["t_idx"] = [comp.tidx(t) for t in expo_times]
                hit["num_atom"] = [a_[0] for a_ in at]
                hit["x_atom"] = [a_[1] for a_ in at]
                hit["expo_row"] = np.arange(E)
                hit["sign"] = 1.0
                expo_tmpl_cache[asset] = hit
            return hit

        # Where the events of a product go depends on its payment dates alone (controller.py:401-426, 451-461): cashflow range
        # [all cash events], then the evaluation range — the exposure events of ALL dates with the cash events spliced in where the
        # reference evaluates them (before the first exposure date that is not earlier), cash events after the last exposure date at
        # the end if cashflows are wanted.  Books of thousands of products have a handful of distinct payment schedules: the layout
        # is computed once per schedule and the event array is filled class by class with array operations (phase B below); the
        # per-product Python work (phase A) is the cash events alone.
        layout_cache: dict = {}

        def layout_of(pdates: tuple):
            hit = layout_cache.get(pdates)
            if hit is None:
                n_cash = len(pdates)
                if want_expo:
                    first = np.searchsorted(expo_arr, np.asarray(pdates, dtype=np.float64), side="left") if n_cash else np.zeros(0, dtype=np.int64)
                    n_sp = int(np.count_nonzero(first < E))                          # (payment dates ascend: the spliced events come first)
                    expo_dest = np.arange(E) + np.searchsorted(first[:n_sp], np.arange(E), side="right")
                    cash_dest = first[:n_sp] + np.arange(n_sp)
                    n_tail = n_cash - n_sp if want_cfs else 0
                    cash_dest = np.concatenate([cash_dest, E + n_sp + np.arange(n_tail)]).astype(np.int64)
                    hit = (expo_dest.astype(np.int64), cash_dest, E + n_sp + n_tail)
                else:
                    hit = (None, np.arange(n_cash, dtype=np.int64), n_cash)
                layout_cache[pdates] = hit
            return hit

        # ---- phase A: per product — cash events (atoms / terms are created in the order the reference visits them) ----------------
        n_prod = len(self.products)
        cash_rows: list = []
        cash_templates: dict = {}
        cash_start = np.zeros(n_prod + 1, dtype=np.int64)
        ev_len = np.zeros(n_prod, dtype=np.int64)
        n_states = np.zeros(n_prod, dtype=np.int32)
        init_state = np.zeros(n_prod, dtype=np.int32)
        cash_classes: dict = {}              # layout -> products
        expo_classes: dict = {}              # (asset, analytic, n_states, layout) -> products
        bs_in: dict = {}                     # analytic exposure inputs of a product
        memo = self.__dict__.setdefault("_pred_memo", {})
        self._member_rec = {}                # product index -> template record (products that took the shortcut below)
        mc_products, mc_set = self._mc_products, self._mc_set
        for p_i, p in enumerate(self.products):
            # Products that differ from an earlier one in strike and sign only (Product._cash_template_key: Europeans on one asset
            # and maturity — 39,400 of the 49,900 products of the reference's PV book) take everything else from that product's
            # record: event rows (the same atoms and the SAME term range), state count, layout, classes, predicates.
            tk = getattr(p, "_cash_template_key", None)
            tkey = tk() if tk is not None else None
            rec = cash_templates.get(tkey) if tkey is not None else None
            if rec is not None and rec["complete"]:
                n_states[p_i], init_state[p_i] = rec["S"], rec["init"]
                for r in rec["rows"]:
                    cash_rows.append(p._cash_template_fill(r))
                cash_start[p_i + 1] = len(cash_rows)
                ev_len[p_i] = rec["ev_len"]
                rec["cash_members"].append(p_i)
                if rec["expo_members"] is not None:
                    rec["expo_members"].append(p_i)
                    if rec["analytic"]:
                        bs_in[p_i] = (float(p._K), float(p._sign())) + rec["bs"]
                mc_products.append(p_i)
                mc_set.add(p_i)
                pid = id(p)
                memo[("skip", pid)], memo[("analytic", pid)], memo[("regression", pid)] = False, rec["analytic"], rec["requires_reg"]
                self._member_rec[p_i] = rec
                continue
            S = p.get_num_states()
            n_states[p_i], init_state[p_i] = S, p.get_initial_state()
            skip = self._can_skip_monte_carlo_for_product(p)
            comp.current_product = p_i
            pdates = self._timelines(p)[0]
            if skip:
                tkey = None
            cash = [] if skip else p._cash_events(comp)
            assert skip or len(cash) == len(pdates)
            first_row = len(cash_rows)
            for ce in cash:
                nt = ce.time if ce.num_time is None else ce.num_time
                num = num_atom_cache.get(nt)
                if num is None:
                    num = num_atom_cache[nt] = comp.atom(AtomicRequest(AtomicRequestType.NUMERAIRE, nt), "numeraire", nt)
                has_x = ce.kind == _abi.EV_EXERCISE or ce.x_time is not None or ce.x_asset is not None       # (an asset id may be None)
                x = comp.atom(AtomicRequest(AtomicRequestType.SPOT), ce.x_asset, ce.time if ce.x_time is None else ce.x_time) if has_x else -1
                co = -1 if ce.reg_idx is None else self._reg_coeff_base[p_i] + ce.reg_idx * S * K
                if ce.coeff_params:
                    co = self._extra_coeff_base[p_i]
                    for k_, v_ in enumerate(ce.coeff_params):
                        comp.coeff_init[co + k_] = float(v_)
                tr = comp.add_terms(ce.terms)
                cash_rows.append((ce.kind, comp.tidx(ce.time), num, x, tr[0], tr[1], co, -1, float(ce.strike), float(ce.sign), tuple(ce.aux)))
            rec = None
            if tkey is not None and all(ce.reg_idx is None and not ce.coeff_params for ce in cash):
                rec = cash_templates[tkey] = dict(rows=cash_rows[first_row:], complete=False)
            cash_start[p_i + 1] = len(cash_rows)
            if skip:
                continue
            lay = layout_of(pdates)
            ev_len[p_i] = lay[2]
            cash_members = cash_classes.setdefault(id(lay), (lay, []))[1]
            cash_members.append(p_i)
            expo_members, analytic = None, False
            if want_expo:
                analytic = self._can_use_analytic_exposure_for_product(p)
                asset = p.asset_ids[0]
                expo_template(asset)                                                  # (creates the asset's exposure atoms HERE, as before)
                expo_members = expo_classes.setdefault((asset, analytic, S, id(lay)), (lay, []))[1]
                expo_members.append(p_i)
                if analytic:
                    _s, sig_p, rate_p = p._bs_inputs(self.model)
                    bs_in[p_i] = (float(p._K), float(p._sign()), float(sig_p), float(rate_p), float(p.exercise_date[0]))
            mc_products.append(p_i)
            mc_set.add(p_i)
            if rec is not None:
                rec.update(complete=True, S=S, init=int(init_state[p_i]), ev_len=lay[2], cash_members=cash_members, expo_members=expo_members,
                           analytic=analytic, bs=bs_in[p_i][2:] if analytic else None, requires_reg=self._product_requires_regression(p))
                self._member_rec[p_i] = rec
        # ---- phase B: the event array, class by class ----------------------------------------------------------------------------
        n_cash = np.diff(cash_start)
        cf_begin = np.concatenate([[0], np.cumsum(n_cash + ev_len)])[:-1]
        ev_begin = cf_begin + n_cash
        n_ev = int((n_cash + ev_len).sum())
        comp.reserve_events(n_ev)
        raw = comp._ev_raw
        crow = (np.array(cash_rows, dtype=_abi.EVENT_DTYPE) if cash_rows else np.zeros(0, dtype=_abi.EVENT_DTYPE)).view(np.float64).reshape(-1, W_EV)
        if len(crow):                                                                 # cashflow ranges: every product's cash events in order
            raw[np.repeat(cf_begin - cash_start[:-1], n_cash) + np.arange(len(crow))] = crow
        for lay, members in cash_classes.values():                                    # the cash events inside the evaluation ranges
            cd = lay[1]
            if len(cd):
                m = np.asarray(members)
                raw[(ev_begin[m][:, None] + cd[None, :]).reshape(-1)] = crow[(cash_start[m][:, None] + np.arange(len(cd))[None, :]).reshape(-1)]
        eb = np.asarray(self._expo_coeff_base, dtype=np.int64)
        for (asset, analytic, S, _l), (lay, members) in expo_classes.items():         # the exposure events
            m = np.asarray(members)
            blk_raw = np.broadcast_to(expo_template(asset).view(np.float64).reshape(1, E, W_EV), (len(m), E, W_EV)).copy()
            blk = blk_raw.reshape(-1).view(_abi.EVENT_DTYPE).reshape(len(m), E)
            if analytic:
                par = np.array([bs_in[q] for q in members])                          # [m][strike, sign, sigma, rate, exercise date]
                blk["kind"] = _abi.EV_EXPO_BS
                blk["coeff_off"] = -1
                blk["strike"], blk["sign"] = par[:, 0:1], par[:, 1:2]
                blk["aux"][:, :, 0], blk["aux"][:, :, 1] = par[:, 2:3], par[:, 3:4]
                blk["aux"][:, :, 2] = par[:, 4:5] - expo_arr[None, :]
            else:
                blk["kind"] = _abi.EV_EXPO_POLY
                blk["coeff_off"] = eb[m][:, None] + np.arange(E, dtype=np.int64)[None, :] * (S * K)
            raw[(ev_begin[m][:, None] + lay[0][None, :]).reshape(-1)] = blk_raw.reshape(-1, W_EV)
        comp.n_events = n_ev
        prods["ev_begin"], prods["ev_end"] = ev_begin, ev_begin + ev_len
        prods["cf_begin"], prods["cf_end"] = cf_begin, ev_begin
        prods["netting_set"] = [self.product_to_netting_set_idx[q] for q in range(n_prod)]
        prods["init_state"], prods["n_states"] = init_state, n_states
        # CVA survival atoms (cva_metric.py:23-46)
        self._cva_atoms = {}
        mt = [float(t) for t in self.metric_exposure_timeline]
        for m_i, m in enumerate(rm.metrics):
            if m.metric_type == MetricType.CVA and m._native:
                surv, cond = [], []
                for k in range(len(mt) - 1):
                    label = (k, m.counterparty_id)
                    surv.append(comp.atom(m.survival_prob_requests[label], m.counterparty_id, mt[k]))
                    cond.append(comp.atom(m.cond_survival_prob_requests[label], m.counterparty_id, mt[k]))
                self._cva_atoms[m_i] = (surv, cond)
        self._comp = comp
        n_state = sum(s.state_dim for s in self.model._slots())
        self._plan_args = (prods, len(self.netting_sets), E if want_expo else 0, off, want_cfs, want_expo, n_state)

    # ---- pre-simulation: Longstaff-Schwartz (controller.py:272-383) ----------------------------------------------
    def _regression_schedule(self, p_i: int, product: Product):
        """backward list of (t_reg, roll_begin, roll_end, store_prod_idx|None, store_expo_idx|None).  Memoised on the
        product's timelines: books of thousands of products share a handful of distinct schedules."""
        rec = self.__dict__.get("_member_rec", {}).get(p_i)          # products of one cash template share their timelines
        if rec is not None and rec.get("sched") is not None:
            return rec["sched"]
        pdates, preg = self._timelines(product)
        cache = self.__dict__.setdefault("_sched_cache", {})
        hit = cache.get((pdates, preg))
        if hit is not None:
            if rec is not None:
                rec["sched"] = hit
            return hit
        reg_tl = sorted(set(preg) | {float(t) for t in self.exposure_timeline})
        P = len(pdates)
        last = P
        sched = []
        pd_arr = np.asarray(pdates)
        reg_pos = {t: i for i, t in enumerate(preg)}
        for t_reg in reversed(reg_tl):
            idx = int(np.searchsorted(pd_arr, t_reg, side="left"))
            if idx >= P:
                continue
            t_next = idx + 1 if pdates[idx] == t_reg else idx
            roll = (t_next, last) if t_next < last else (last, last)
            if t_next < last:
                last = t_next
            sched.append((t_reg, roll[0], roll[1], reg_pos.get(t_reg), self._exposure_time_to_idx.get(t_reg)))
        cache[(pdates, preg)] = sched
        if rec is not None:
            rec["sched"] = sched
        return sched

    def _regression_atoms(self, sched, asset_id):
        """(numeraire atom, explanatory SPOT atom) of every step of a schedule, memoised per (schedule, asset)"""
        cache = self.__dict__.setdefault("_sched_atom_cache", {})
        key = (id(sched), asset_id)
        hit = cache.get(key)
        if hit is None:
            comp = self._comp
            hit = [(comp.atom(AtomicRequest(AtomicRequestType.NUMERAIRE, t_reg), "numeraire", t_reg),
                    comp.atom(AtomicRequest(AtomicRequestType.SPOT), asset_id, t_reg)) for (t_reg, *_r) in sched]
            cache[key] = hit
        return hit

    def _perform_regression(self, shard: Shard, sim_plan: SimPlan, sim):
        be = self.backend
        off, n_local = shard.split(self.num_paths_presim)
        eng = MonteCarloEngine(self.simulation_timeline, self.simulation_scheme, self.model, n_local, self.num_steps,
                               is_pre_simulation=True, path_offset=off, backend=be, plan=sim_plan, sim=sim,
                               seed_offset=self.seed_offset)
        if "pre" in self._inject:
            eng.inject_z, eng.inject_u = self._inject["pre"]
        paths = eng.generate_paths_native()
        self.last_state["paths_pre"] = paths
        K = self.regression_function.get_degree()
        comp = self._comp
        jobs = []
        for p_i, p in enumerate(self.products):
            if p_i not in self._mc_set or not self._product_requires_regression(p):
                continue
            sched = self._regression_schedule(p_i, p)
            jobs.append((p_i, p, sched, self._regression_atoms(sched, p.asset_ids[0])))
        self._book_ready()               # (everything above ran beside a big book's upload)
        self._collapse_values(shard, paths)
        self._set_bridge_rng(eng.seed, off, "bridge_pre")
        self._set_exercise_replay("pre", n_local)
        if len(self._comp.atoms) != len(self.book_plan.atoms):
            raise RuntimeError("internal: regression atoms must be registered before the book is frozen")
        # range of every explanatory variable (conditioning of the monomial basis; exact-degeneracy detection)
        x_ids = sorted({x for _, _, _, atoms in jobs for _, x in atoms})
        if x_ids:
            mm = be.lsm_stats(self.book, x_ids, paths)
            g = shard.all_gather_np(mm)
            lo, hi = g[:, :, 0].min(axis=0), g[:, :, 1].max(axis=0)
            x_range = {x: (lo[i], hi[i]) for i, x in enumerate(x_ids)}
        lsm_flags = (_abi.LSM_MFMA if self.use_mfma else 0) | (_abi.LSM_F32_CACHE if self.reference_float32_cf_cache else 0)
        if not jobs:
            return
        if self.batch_lsm and not self.use_mfma and len(jobs) >= 4:       # (one or two products: the per-product loop has less fixed cost per step)
            return self._perform_regression_batched(shard, jobs, x_range, paths, n_local, K, lsm_flags)
        for p_i, p, sched, atoms in jobs:
            S = p.get_num_states()
            W = be.zeros(S, n_local)
            if hasattr(be, "lsm_run") and (shard.world == 1 or shard.device_collectives) \
                    and self._lsm_on_device(be, p_i, p, sched, atoms, x_range, paths, W, S, K, lsm_flags, shard):
                continue
            W.zero_()
            for (t_reg, r0, r1, prod_idx, expo_idx), (num, x) in zip(sched, atoms):
                xmin, xmax = x_range[x]
                degenerate = not (xmax > xmin)
                shift = 0.5 * (xmin + xmax) if not degenerate else xmin
                scale = 2.0 / (xmax - xmin) if not degenerate else 1.0
                mom = be.lsm_step(self.book, p_i, r0, r1, num, x, shift, scale, paths, W, flags=lsm_flags)
                shard.all_reduce_(mom)
                coeffs = solve_normal_equations(mom.detach().cpu().numpy(), K, S, shift, scale, degenerate, xmin)
                if prod_idx is not None:
                    p.regression_coeffs[prod_idx] = torch.from_numpy(coeffs)
                    be.book_set_coeffs(self.book, self._reg_coeff_base[p_i] + prod_idx * S * K, coeffs)
                if expo_idx is not None:
                    self.regression_coeffs[p_i][expo_idx] = torch.from_numpy(coeffs)
                    be.book_set_coeffs(self.book, self._expo_coeff_base[p_i] + expo_idx * S * K, coeffs)

    def _collapse_values(self, shard: Shard, paths):
        """events that sum many atoms of one state variable (a Bermudan swaption's exercise value: the underlying swap priced
        from ~35-64 zero-bond requests per date, bermudan_option.py:40-43) become one host-verified polynomial of that variable
        on the range the pre-simulation paths visit, padded by 30 % on either side (mcx_book_collapse_values); the LSM roll and
        the main pass evaluate ~20 multiply-adds per path and date instead of ~15 instructions per term, paths outside the
        verified range take the exact term loop.  All ranks use the same (gathered) ranges: results do not depend on the sharding."""
        be = self.backend
        if not self.collapse_values or not hasattr(be, "book_collapse_values"):
            return
        ev = self.book_plan.events
        if len(ev) == 0 or int((ev["term_end"] - ev["term_begin"]).max()) < self.collapse_min_terms:
            return
        mm = be.rows_minmax(paths)                                     # [T * D][2]
        g = shard.all_gather_np(mm)
        T, D = paths.shape[0], paths.shape[1]
        lo, hi = g[:, :, 0].min(axis=0).reshape(T, D), g[:, :, 1].max(axis=0).reshape(T, D)
        self.n_collapsed_events = be.book_collapse_values(self.book, lo, hi, self.collapse_pad, self.collapse_rel_tol, self.collapse_min_terms)

    def _lsm_on_device(self, be, p_i, p, sched, atoms, x_range, paths, W, S, K, lsm_flags, shard=None) -> bool:
        """the product's whole backward induction enqueued on the device (mcx_lsm_run): roll, moments, K x K solve and coefficient
        scatter of every date back to back, one synchronisation at the end.  False when a system came out numerically singular
        (the caller then runs the per-date loop with the host solver, which falls back to lstsq)."""
        dates = np.zeros(len(sched), dtype=_abi.LSM_DATE_DTYPE)
        for j, ((t_reg, r0, r1, prod_idx, expo_idx), (num, x)) in enumerate(zip(sched, atoms)):
            xmin, xmax = x_range[x]
            degenerate = not (xmax > xmin)
            d = dates[j]
            d["roll_begin"], d["roll_end"], d["num_atom"], d["x_atom"], d["degenerate"] = r0, r1, num, x, int(degenerate)
            d["shift"] = 0.5 * (xmin + xmax) if not degenerate else xmin
            d["scale"] = 2.0 / (xmax - xmin) if not degenerate else 1.0
            d["x0"] = xmin
            d["coeff_off"][0] = -1 if prod_idx is None else self._reg_coeff_base[p_i] + prod_idx * S * K
            d["coeff_off"][1] = -1 if expo_idx is None else self._expo_coeff_base[p_i] + expo_idx * S * K
        if shard is not None and shard.device_collectives:
            # several GPUs: roll + moments of this rank's pre-simulation paths, all-reduce over RCCL, solve — per date, all
            # stream-ordered on the device (<= 120 latency-bound all-reduces of ~15 doubles, no host hop in between)
            tab = be.zeros(len(dates) * S * K)
            st = be.zeros(len(dates), dtype=torch.int32)
            for j in range(len(dates)):
                d = dates[j]
                mom = be.lsm_step(self.book, p_i, int(d["roll_begin"]), int(d["roll_end"]), int(d["num_atom"]), int(d["x_atom"]),
                                  float(d["shift"]), float(d["scale"]), paths, W, flags=lsm_flags)
                shard.all_reduce_(mom)
                be.lsm_solve(self.book, p_i, mom, dates[j:j + 1], j, tab, st)
            coeffs, status = tab.cpu().numpy().reshape(len(dates), S, K), st.cpu().numpy()
        else:
            coeffs, status = be.lsm_run(self.book, p_i, dates, paths, W, flags=lsm_flags)
        if status.any():
            return False
        for j, (t_reg, r0, r1, prod_idx, expo_idx) in enumerate(sched):
            if prod_idx is not None:
                p.regression_coeffs[prod_idx] = torch.from_numpy(coeffs[j].copy())
            if expo_idx is not None:
                self.regression_coeffs[p_i][expo_idx] = torch.from_numpy(coeffs[j].copy())
        return True

    def _perform_regression_batched(self, shard: Shard, jobs, x_range, paths, n_local: int, K: int, lsm_flags: int):
        """The backward LSM induction of ALL products at once: step r of every product's schedule runs in one launch per
        exercise-state count (mcx_lsm_step_batch), the K x K systems are solved in one batched numpy call and the
        coefficients go back in one scatter.  Same arithmetic per (product, date) as the per-product loop above
        (reference: controller.py:289-383, one Python iteration per product and date)."""
        be = self.backend
        S_of = [p.get_num_states() for _, p, _, _ in jobs]
        S_arr = np.asarray(S_of, dtype=np.int64)
        w_off = np.concatenate([[0], np.cumsum(S_arr * n_local)])
        tot = int(w_off[-1])
        W = be.zeros(max(tot, 1))
        mirror = np.zeros(len(self.book_plan.coeffs))                 # host image of the coefficients this pass produces
        # products sharing (schedule, explanatory asset, state count) differ only in their ids and offsets: one vectorised
        # block of the job table per class and step instead of a Python iteration per (product, date)
        pi_all = np.fromiter((job[0] for job in jobs), dtype=np.int64, count=len(jobs))
        reg_all = np.asarray(self._reg_coeff_base, dtype=np.int64)[pi_all]
        expo_all = np.asarray(self._expo_coeff_base, dtype=np.int64)[pi_all]
        classes: dict = {}
        for j, (p_i, p, sched, atoms) in enumerate(jobs):
            classes.setdefault((id(sched), id(atoms), S_of[j]), []).append(j)
        cls = []
        for (_, _, S), members in classes.items():
            _, _, sched, atoms = jobs[members[0]]
            L = len(sched)
            lo = np.array([x_range[x][0] for _, x in atoms]); hi = np.array([x_range[x][1] for _, x in atoms])
            deg = ~(hi > lo)
            m_ = np.asarray(members)
            cls.append(dict(S=S, L=L, members=m_, p_i=pi_all[m_].astype(np.int32), w_off=w_off[m_],
                            reg_base=reg_all[m_], expo_base=expo_all[m_],
                            r0=[s[1] for s in sched], r1=[s[2] for s in sched],
                            prod_idx=[-1 if s[3] is None else s[3] for s in sched],
                            expo_idx=[-1 if s[4] is None else s[4] for s in sched],
                            num=[a_[0] for a_ in atoms], x=[a_[1] for a_ in atoms], xmin=lo, deg=deg,
                            shift=np.where(deg, lo, 0.5 * (lo + hi)), scale=np.where(deg, 1.0, 2.0 / np.where(deg, 1.0, hi - lo))))
        # the job / solve tables of ALL steps at once.  Step order as the reference walks it (one date further back per step, the
        # products of one exercise-state count per launch): step (r, S) holds, class by class, the members of every class with S
        # states whose schedule is longer than r.  Destination rows by cumulative sums over a [class][r] matrix, fields by
        # broadcast assignment per class — no Python iteration per (product, date) or per step.
        n_cls, max_len = len(cls), max(cl["L"] for cl in cls)
        L_c = np.array([cl["L"] for cl in cls]); S_c = np.array([cl["S"] for cl in cls]); n_c = np.array([len(cl["members"]) for cl in cls])
        S_vals = sorted(set(S_c.tolist()))
        active = np.arange(max_len)[None, :] < L_c[:, None]
        within = np.zeros((n_cls, max_len), dtype=np.int64)
        size = np.zeros((max_len, len(S_vals)), dtype=np.int64)
        for k_, S in enumerate(S_vals):
            sz = n_c[:, None] * (active & (S_c == S)[:, None])
            within += np.where((S_c == S)[:, None], np.cumsum(sz, axis=0) - sz, 0)
            size[:, k_] = sz.sum(axis=0)
        flat = size.reshape(-1)
        start = np.concatenate([[0], np.cumsum(flat)])
        keep = flat > 0
        step_states = np.tile(np.array(S_vals, dtype=np.int32), max_len)[keep]
        step_begin = np.concatenate([[0], np.cumsum(flat[keep])]).astype(np.int32)
        n_jobs = int(start[-1])
        jt = np.zeros(n_jobs, dtype=_abi.LSM_JOB_DTYPE)
        sj = np.zeros(n_jobs, dtype=_abi.LSM_SOLVE_JOB_DTYPE)
        sj_off = sj["coeff_off"]
        for c_, cl in enumerate(cls):
            S, L, n_m = cl["S"], cl["L"], int(n_c[c_])
            d0 = start[np.arange(L) * len(S_vals) + S_vals.index(S)] + within[c_, :L]
            dest = (d0[:, None] + np.arange(n_m)[None, :]).reshape(-1)
            col = lambda v, dt=None: np.repeat(np.asarray(v, dtype=dt), n_m)            # one value per step -> every member
            row = lambda v: np.tile(v, L)                                              # one value per member -> every step
            jt["product"][dest], jt["w_offset"][dest] = row(cl["p_i"]), row(cl["w_off"])
            jt["roll_begin"][dest], jt["roll_end"][dest] = col(cl["r0"]), col(cl["r1"])
            jt["num_atom"][dest], jt["x_atom"][dest] = col(cl["num"]), col(cl["x"])
            jt["shift"][dest] = sj["shift"][dest] = col(cl["shift"])
            jt["scale"][dest] = sj["scale"][dest] = col(cl["scale"])
            sj["x0"][dest], sj["degenerate"][dest] = col(cl["xmin"]), col(cl["deg"], np.int32)
            pr, ex = np.asarray(cl["prod_idx"], dtype=np.int64), np.asarray(cl["expo_idx"], dtype=np.int64)
            t_prod = np.where(pr[:, None] >= 0, cl["reg_base"][None, :] + pr[:, None] * (S * K), -1)
            t_expo = np.where(ex[:, None] >= 0, cl["expo_base"][None, :] + ex[:, None] * (S * K), -1)
            has_p = np.broadcast_to(pr[:, None] >= 0, t_prod.shape)                   # at most two targets per system, the product block first
            sj_off[dest, 0] = np.where(has_p, t_prod, t_expo).reshape(-1)
            sj_off[dest, 1] = np.where(has_p, t_expo, -1).reshape(-1)
        # solves on the device when the backend has them; a singular system anywhere repeats the induction with the host solver
        # (which falls back to lstsq)
        on_device = hasattr(be, "lsm_solve_batch") and not getattr(self, "_lsm_host_solves", False)
        singular = 0
        if on_device and shard.world == 1 and hasattr(be, "lsm_run_batch") and getattr(self, "lsm_one_call", True):
            # one GPU: every (step, solve) pair enqueued by ONE library call (mcx_lsm_run_batch), tables uploaded once
            singular = be.lsm_run_batch(self.book, jt, sj, step_begin, step_states, paths, W, n_local, flags=lsm_flags)
        else:
            flag = be.zeros(1, dtype=torch.int32) if on_device else None
            for t_ in range(len(step_states)):
                j0, j1, S = int(step_begin[t_]), int(step_begin[t_ + 1]), int(step_states[t_])
                arr, sv = jt[j0:j1], sj[j0:j1]
                if on_device:
                    # several GPUs under torch.distributed: (step, all-reduce, solve) enqueued back to back, stream-ordered
                    mom = be.lsm_step_batch_dev(self.book, arr, S, paths, W, n_local, flags=lsm_flags)
                    shard.all_reduce_(mom)
                    be.lsm_solve_batch(self.book, sv, S, mom, flag)
                    continue
                mom = be.lsm_step_batch(self.book, arr, S, paths, W, n_local, flags=lsm_flags)
                mom = shard.all_reduce_np(mom)
                coeffs = solve_normal_equations_batch(mom, K, S, arr["shift"], arr["scale"], sv["degenerate"].astype(bool),
                                                      sv["x0"]).reshape(len(arr), S * K)
                tgt = sv["coeff_off"]
                rows = np.concatenate([np.nonzero(tgt[:, w] >= 0)[0] for w in (0, 1)])
                if len(rows):
                    offs = np.concatenate([tgt[tgt[:, w] >= 0, w] for w in (0, 1)]).astype(np.int64)
                    vals = coeffs[rows]
                    be.book_set_coeffs_batch(self.book, offs, vals)
                    mirror[offs[:, None] + np.arange(S * K)[None, :]] = vals
            if on_device:
                singular = int(flag.cpu()[0])
        if on_device:
            if singular != 0:
                self.lsm_singular_retries = getattr(self, "lsm_singular_retries", 0) + 1
                self._lsm_host_solves = True
                try:
                    be.book_reset_coeffs(self.book, self._coeffs_at_upload)
                    return self._perform_regression_batched(shard, jobs, x_range, paths, n_local, K, lsm_flags)
                finally:
                    self._lsm_host_solves = False
            mirror = be.book_get_coeffs(self.book)
            self.book.plan.coeffs[:] = mirror                                 # (the host image of the uploaded book follows)
        E = len(self.exposure_timeline)
        mt = torch.from_numpy(mirror)                                 # the products' coefficient tensors are views of this one image
        for j, (p_i, p, _, _) in enumerate(jobs):
            S = S_of[j]
            b0 = self._expo_coeff_base[p_i]
            self.regression_coeffs[p_i] = mt[b0:b0 + E * S * K].view(E, S, K)
            R = len(self._timelines(p)[1])
            if R:
                b1 = self._reg_coeff_base[p_i]
                p.regression_coeffs = mt[b1:b1 + R * S * K].view(R, S, K)

    def _set_bridge_rng(self, seed: int, path_offset: int, inject_key: str):
        """Brownian-bridge barrier events draw their uniforms inside the book / LSM kernels (mcx_book_set_bridge_rng)"""
        if any(getattr(p, "use_brownian_bridge", False) for p in self.products):
            self.backend.book_set_bridge_rng(self.book, int(seed), int(path_offset), self._inject.get(inject_key))

    def _set_exercise_replay(self, phase: str, n_paths: int):
        """exercise decisions of this phase ("pre" / "main") recorded into, or replayed from, `self.exercise_replay`
        (mcx/aad.py run_with_bumps: the bumped runs of an exercise product keep the base run's exercise policy, as the
        reference's tape does — bermudan_option.py:122-128 puts no gradient through `should_exercise`)"""
        rp = getattr(self, "exercise_replay", None)
        be = self.backend
        if not rp:
            # a book cached across runs (_compile_all) must not keep the mode / buffer of an earlier replayed run
            if getattr(self, "_replay_set", False):
                be.book_set_exercise_replay(self.book, 0, None)
                self._replay_set = False
            return
        self._replay_set = True
        if rp["mode"] == 1 and rp.get(phase) is None:
            rp[phase] = be.new_exercise_bits(len(self.book_plan.events), n_paths)
        be.book_set_exercise_replay(self.book, rp["mode"], rp[phase])

    def _register_regression_atoms(self):
        """atoms the LSM needs must exist before the book is uploaded"""
        for p_i, p in enumerate(self.products):
            if p_i in self._mc_set and self._product_requires_regression(p):
                self._regression_atoms(self._regression_schedule(p_i, p), p.asset_ids[0])

    def perform_prepocessing(self, request_interface: RequestInterface):   # (sic) reference spelling
        request_interface.collect_and_index_requests(self.products, self.simulation_timeline, self._get_requests(),
                                                     self.metric_exposure_timeline)

    # ---- main simulation: metrics (controller.py:506-563) ----------------------------------------------------------
    def _zero_metric_result(self, metric: Metric):
        n = 1 if metric.metric_type in {MetricType.PV, MetricType.CVA, MetricType.EEPE} else len(self.metric_exposure_timeline)
        return [(0.0, 0.0) for _ in range(n)]

    def _radix_select(self, shard: Shard, E: int, n_sel: int, rem, hist_pass, digits=_SELECT_DIGITS, keys=False) -> np.ndarray:
        """the six digit passes of an exact select, enqueued back to back: histogram (hist_pass), all-reduce over the ranks
        (stream-ordered), bin selection on the device; ONE copy of the final keys (a host round trip per pass is six stalls of the
        stream).  rem: int64 [E][n_sel] device tensor of ranks (consumed).  shard None: a rank-local select, no collective.
        digits: a leading subset of the passes leaves the low bits of the keys zero (keys=True returns the uint64 keys)."""
        be = self.backend
        prefix = be.zeros(E, n_sel, dtype=torch.int64)
        buf = self._buffer_typed("select_hist", torch.int64, E * n_sel * (1 << max(b for _, b in _SELECT_DIGITS)))
        for shift, bits in digits:
            hist = buf[:E * n_sel * (1 << bits)].view(E, n_sel, 1 << bits)
            hist_pass(prefix, shift, bits, hist)
            if shard is not None:
                shard.all_reduce_(hist)
            be.select_narrow(hist, E, n_sel, shift, bits, prefix, rem)
        k = prefix.cpu().numpy().view(np.uint64)
        return k if keys else _key_to_double(k)

    def _select_order_stats(self, shard: Shard, unsec: UnsecuredSpec, expo_ns, ranks: list[int]) -> np.ndarray:
        """exact global order statistics x_(r) for r in ranks at every metric date -> [n_dates][len(ranks)]"""
        be = self.backend
        E, n_sel = unsec.n_dates, len(ranks)
        if hasattr(be, "select_narrow"):
            if hasattr(be, "select_bracket") and self.bracket_select:
                out = self._select_with_bracket(shard, unsec, expo_ns, ranks)
                if out is not None:
                    return out
            rem = be.from_numpy(np.tile(np.asarray(ranks, dtype=np.int64), (E, 1)))
            return self._radix_select(shard, E, n_sel, rem, lambda prefix, shift, bits, hist:
                                      be.select_hist_dev(unsec, expo_ns, n_sel, prefix, shift, bits, hist))
        prefix = np.zeros((E, n_sel), dtype=np.uint64)
        rem = np.tile(np.asarray(ranks, dtype=np.int64), (E, 1))
        for shift, bits in _SELECT_DIGITS:
            hist = be.select_hist(unsec, expo_ns, n_sel, prefix, shift, bits)
            shard.all_reduce_(hist)
            cum = np.cumsum(hist.cpu().numpy(), axis=-1)
            b = (cum <= rem[..., None]).sum(axis=-1)
            below = np.where(b > 0, np.take_along_axis(cum, np.maximum(b - 1, 0)[..., None], axis=-1)[..., 0], 0)
            rem = rem - below
            prefix = prefix | (b.astype(np.uint64) << np.uint64(shift))
        return _key_to_double(prefix)

    def _select_with_bracket(self, shard: Shard, unsec: UnsecuredSpec, expo_ns, ranks: list[int]):
        """The select in ONE pass over the exposure matrix instead of six (pfe_metric.py:49-73 sorts it).  Per date the wanted order
        statistics lie, almost surely, between two order statistics of a sample of the paths: (1) exact select of the bracket
        ends on the first `bracket_sample` local paths (paths are exchangeable; the widest bracket over the ranks is taken);
        (2) mcx_select_bracket counts the paths below the bracket and gathers the ~1-2 % inside it; (3) the digit passes run on the
        gathered candidates.  Exactness never rests on the bracket: a date whose ranks are not inside it (checked on the global
        counts) or whose candidates overflowed their buffer — every path at one value, say — is redone by the digit passes over
        the matrix.  None: too few paths for the detour to pay."""
        be = self.backend
        E, n_sel = unsec.n_dates, len(ranks)
        n_local, n_tot = int(expo_ns.shape[1]), int(self.num_paths_mainsim)
        S = min(n_local, int(self.bracket_sample))
        if n_tot < 32 * self.bracket_sample or S < 4096:
            return None
        r_min, r_max = min(ranks), max(ranks)
        p = 0.5 * (r_min + r_max) / n_tot
        sig = math.sqrt(S * max(p * (1.0 - p), 1.0 / S))
        s_lo, s_hi = int(math.floor(r_min / n_tot * S - 5.0 * sig - 2)), int(math.ceil(r_max / n_tot * S + 5.0 * sig + 2))
        open_lo, open_hi = s_lo < 0, s_hi > S - 1
        s_rk = [min(max(s_lo, 0), S - 1), min(max(s_hi, 0), S - 1)]
        rem_s = be.from_numpy(np.tile(np.asarray(s_rk, dtype=np.int64), (E, 1)))
        # (exact order statistics of the sample: a date on which the sampled paths share one exposure then gives lo == hi)
        ends = self._radix_select(None, E, 2, rem_s, lambda prefix, shift, bits, hist:
                                  be.select_hist_dev(unsec, expo_ns, 2, prefix, shift, bits, hist, n_paths=S)).copy()
        if open_lo:
            ends[:, 0] = -np.inf
        if open_hi:
            ends[:, 1] = np.inf
        g = shard.all_gather_np(ends)                                      # [world][E][2]: every rank sampled its own paths
        lo, hi = g[:, :, 0].min(axis=0), g[:, :, 1].max(axis=0)
        width = (s_rk[1] - s_rk[0] + 1) / S                                # expected share of the paths inside the bracket
        cap = int(min(n_local, max(8192, 4.0 * width * n_local)))
        cap = (cap + 1023) // 1024 * 1024
        cand = self._buffer("select_cand", E, cap)
        counts = self._buffer_typed("select_counts", torch.int64, 2 * E)[:2 * E].view(2, E)
        be.select_bracket(unsec, expo_ns, be.from_numpy(lo), be.from_numpy(hi), counts, cand)
        local_n = counts[1].clone()                                          # candidates this rank holds (before the sum over ranks)
        shard.all_reduce_(counts)
        cnt = counts.cpu().numpy()
        lost_bit = np.int64(_abi.SELECT_LOST)
        below, inside = cnt[0], cnt[1] & (lost_bit - 1)
        local_cnt = local_n.cpu().numpy() & (lost_bit - 1)
        # some rank's candidate row overflowed, or some wave could not stage all its candidates (MCX_SELECT_LOST)
        over = (shard.all_reduce_np((local_cnt > cap).astype(np.float64)) > 0) | (cnt[1] >= lost_bit)
        contained = (below <= r_min) & (r_max < below + inside)
        # a bracket that is one point (every sampled path at one exposure: before the first date, after the last exercise, the
        # threshold's atom at zero): everything inside it IS that value, the candidates are not needed
        point = contained & (lo == hi)
        ok = contained & ~over & ~point
        out = np.zeros((E, n_sel))
        out[point] = lo[point][:, None]
        if ok.any():
            rem = np.asarray(ranks, dtype=np.int64)[None, :] - below[:, None]
            rem[~ok] = 0
            row_n = be.from_numpy(np.where(ok, np.minimum(local_cnt, cap), 0).astype(np.int64))
            vals = self._radix_select(shard, E, n_sel, be.from_numpy(rem), lambda prefix, shift, bits, hist:
                                      be.select_hist_rows(cand, row_n, n_sel, prefix, shift, bits, hist))
            out[ok] = vals[ok]
        bad = np.nonzero(~ok & ~point)[0]
        if len(bad):
            sub = UnsecuredSpec(unsec.rows[bad], None if unsec.delayed is None else unsec.delayed[bad], float(unsec.desc.threshold),
                                bool(unsec.desc.collateralized))
            rem_b = be.from_numpy(np.tile(np.asarray(ranks, dtype=np.int64), (len(bad), 1)))
            out[bad] = self._radix_select(shard, len(bad), n_sel, rem_b, lambda prefix, shift, bits, hist:
                                          be.select_hist_dev(sub, expo_ns, n_sel, prefix, shift, bits, hist))
        self.last_select = dict(bracket_dates=int(ok.sum()), point_dates=int(point.sum()), fallback_dates=int(len(bad)), cap=cap, sample=S)
        return out

    def _evaluate_netting_set(self, shard: Shard, ns_i: int, ns: NettingSet, cfs, expo, paths, has_pathwise: bool,
                              analytical_acc: list[float], fr: dict | None = None):
        be, rm = self.backend, self.risk_metrics
        n_total = self.num_paths_mainsim
        want_expo = rm.requires_exposure_profiles()
        unsec = None
        expo_ns = None
        if want_expo:
            delayed = self.netting_set_delayed_exposure_indices[ns_i].numpy() if ns.is_collateralized() else None
            unsec = UnsecuredSpec(self.metric_exposure_indices.numpy(), delayed, ns.threshold, ns.is_collateralized())
            expo_ns = expo[ns_i] if expo is not None else None
        prof = fr.get("prof") if fr else None

        def profiles():
            nonlocal prof
            if prof is None:
                local = be.reduce_profiles(unsec, expo_ns).view(np.float64).reshape(unsec.n_dates, 2, 4)
                prof = shard.all_gather_np(local)          # [world][E][2][4]
            return prof

        def pv_records():
            if fr and fr.get("pv") is not None:
                return fr["pv"]
            return shard.all_gather_np(be.reduce_vector(cfs[ns_i]).view(np.float64))

        out = []
        for m_i, metric in enumerate(rm.metrics):
            mt = metric.metric_type
            if mt == MetricType.CVA and ns.counterparty_id is not None \
                    and getattr(metric, "counterparty_id", None) != ns.counterparty_id:
                out.append(self._zero_metric_result(metric))                           # controller.py:536-542
                continue
            if mt == MetricType.PV and metric.evaluation_type == Metric.EvaluationType.ANALYTICAL:
                val, err = 0.0, 0.0
                if has_pathwise:
                    val, err = mean_and_error(pv_records())
                out.append([(analytical_acc[m_i] + val, err)])
                continue
            if not metric._native:
                out.append(self._evaluate_plugin_metric(metric, ns, unsec, expo_ns, cfs, ns_i, paths))
                continue
            if mt == MetricType.PV:
                out.append([mean_and_error(pv_records())])
            elif mt == MetricType.EPE:
                out.append([mean_and_error(profiles()[:, m, 0]) for m in range(unsec.n_dates)])
            elif mt == MetricType.ENE:
                out.append([mean_and_error(profiles()[:, m, 1]) for m in range(unsec.n_dates)])
            elif mt == MetricType.CE:
                out.append([mean_and_error(profiles()[:, 0, 0])])
            elif mt == MetricType.EEPE:
                ee = np.array([mean_and_error(profiles()[:, m, 0])[0] for m in range(unsec.n_dates)])
                err = float(np.std(ee, ddof=1) / math.sqrt(len(ee))) if len(ee) > 1 else float("nan")
                out.append([(float(ee.mean()), err)])
            elif mt == MetricType.CVA:
                if fr and fr.get("cva") is not None and fr.get("cva_metric") == m_i:
                    out.append([mean_and_error(fr["cva"])])
                else:
                    surv, cond = self._cva_atoms[m_i]
                    rec = be.reduce_cva(self.book, unsec, surv, cond, metric.recovery_rate, expo_ns, paths)
                    out.append([mean_and_error(shard.all_gather_np(rec.view(np.float64)))])
            elif mt == MetricType.PFE:
                q = metric.q_index(n_total)
                edge = q == 0 or q == n_total - 1
                ranks = [q] if edge else [q - 1, q, q + 1]
                vals = self._select_order_stats(shard, unsec, expo_ns, ranks)
                res = []
                for m in range(unsec.n_dates):
                    if edge:
                        res.append((float(vals[m, 0]), 0.0))
                    else:
                        lo, mid, hi = (float(v) for v in vals[m])
                        res.append((mid, metric.quantile_error(lo, mid, hi, q, n_total)))
                out.append(res)
            else:
                raise NotImplementedError(mt)
        return out

    def _evaluate_plugin_metric(self, metric, ns, unsec, expo_ns, cfs, ns_i, paths):
        """user-defined Metric subclass: hand it torch views of the device buffers (Metrics API, metric.py:37-60)"""
        be = self.backend
        exposures = []
        if unsec is not None:
            u = be.unsecured(unsec, expo_ns)
            exposures = [u[m] for m in range(unsec.n_dates)]
        cf = cfs[ns_i] if cfs is not None else be.zeros(paths.shape[2])
        ri = RequestInterface(self.model)
        self.perform_prepocessing(ri)
        resolved = ri.resolve_requests(paths.permute(2, 0, 1))
        res = metric.evaluate(exposures=exposures, cfs=cf, resolved_requests=resolved, netting_set=ns, model=self.model)
        return [(float(v), float(e)) for v, e in res]

    # the path shard of this process: torch.distributed's default group (mcx/parallel.py).  Tests replace the factory by an
    # in-process stand-in that emulates several ranks on one GPU (tests/emulated_ranks.py).
    shard_factory = Shard

    # ---- entry point (controller.py:663-709) ------------------------------------------------------------------------
    @single_threaded_host()
    def prepare(self):
        """everything before the main simulation: descriptor compilation + (if needed) pre-simulation and LSM regression
        (the reference's perform_prepocessing, controller.py:257-292)"""
        be = self.backend
        self._shard = self.shard_factory()
        self.last_state = {}             # release the previous run's device buffers first: the allocator can reuse them
        t0 = time.perf_counter()
        self._compile_all()
        t1 = time.perf_counter()
        self.sim_plan = SimPlan(self.model, self.simulation_timeline.numpy(), self.simulation_scheme, self.num_steps)
        self._sim = be.sim_create(self.sim_plan)
        if self.requires_regression:
            self._perform_regression(self._shard, self.sim_plan, self._sim)
        self._book_ready()
        self.prepare_timings = dict(compile=t1 - t0, presim_and_regression=time.perf_counter() - t1)
        off, n_local = self._shard.split(self.num_paths_mainsim)
        self._main_engine = MonteCarloEngine(self.simulation_timeline, self.simulation_scheme, self.model, n_local,
                                             self.num_steps, is_pre_simulation=False, path_offset=off, backend=be,
                                             plan=self.sim_plan, sim=self._sim, seed_offset=self.seed_offset)
        if "main" in self._inject:
            self._main_engine.inject_z, self._main_engine.inject_u = self._inject["main"]
        replaying = bool(getattr(self, "exercise_replay", None))
        self._fused = self._build_fused() if (self.allow_fused and self._mc_products and not replaying) else None
        if replaying:
            self._set_exercise_replay("main", n_local)
        self._set_bridge_rng(self._main_engine.seed, self._main_engine.path_offset, "bridge_main")     # after the pre-simulation's
        be.synchronize()

    def _build_fused(self):
        """FusedPlan for the one-launch main pass, or None when some requested metric needs the separate kernels"""
        rm = self.risk_metrics
        if len(self.netting_sets) > _abi.FUSED_MAX_NS or any(ns.is_collateralized() for ns in self.netting_sets):
            return None
        if not all(m._native for m in rm.metrics):
            return None
        want_expo = rm.requires_exposure_profiles()
        kinds = {m.metric_type for m in rm.metrics}
        want_prof = bool(kinds & {MetricType.EPE, MetricType.ENE, MetricType.CE, MetricType.EEPE})
        self._fused_needs_expo = MetricType.PFE in kinds
        specs, self._fused_cva_metric = [], []
        for ns_i, ns in enumerate(self.netting_sets):
            cva = [m_i for m_i, m in enumerate(rm.metrics) if m.metric_type == MetricType.CVA
                   and not (ns.counterparty_id is not None and m.counterparty_id != ns.counterparty_id)]
            if len(cva) > 1:
                return None
            sp = dict(rows=self.metric_exposure_indices.numpy() if want_expo else np.zeros(0, dtype=np.int32),
                      want_profiles=want_prof and want_expo, threshold=ns.threshold, surv=None, cond=None, recovery=0.0)
            if cva:
                sp["surv"], sp["cond"] = self._cva_atoms[cva[0]]
                sp["recovery"] = rm.metrics[cva[0]].recovery_rate
            self._fused_cva_metric.append(cva[0] if cva else None)
            specs.append(sp)
        tl = {float(t): i for i, t in enumerate(self.simulation_timeline)}
        row_t = [tl[float(t)] for t in self.exposure_timeline] if want_expo else []
        plan = FusedPlan(specs, rm.any_pv, row_t)
        if plan.n_records == 0:
            return None
        return self.backend.fused_create(self._sim, self.book, plan)

    def _buffer(self, name: str, *shape):
        """device buffer owned by the controller and reused by every later pass of the same shape (a 2 M-path exposure matrix is
        2 GB: re-allocating it per run leaves the caching allocator to split and re-grow its segments, tens of ms per run)"""
        bufs = self.__dict__.setdefault("_buffers", {})
        t = bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            bufs[name] = None
            # (the path / exposure / cashflow matrices of a run without injected draws take the backend's padded leading dimension)
            padded = name in ("paths", "expo", "cfs") and hasattr(self.backend, "empty_padded") and not self._inject
            t = bufs[name] = self.backend.empty_padded(*shape) if padded else self.backend.empty(*shape)
        return t

    def release_device_buffers(self):
        """drop the device tensors this controller holds (paths, exposures, LSM cache, pipeline buffers): they go back to torch's
        caching allocator at once instead of when the garbage collector gets to the controller (a bumped controller of 1 M paths
        pins ~200 MB that the next one would otherwise have to hipMalloc again: ~80 ms)"""
        self.last_state = {}
        self.__dict__.pop("_buffers", None)
        self.__dict__.pop("_pipe", None)

    def _buffer_typed(self, name: str, dtype, numel: int):
        bufs = self.__dict__.setdefault("_buffers", {})
        t = bufs.get(name)
        if t is None or t.dtype != dtype or t.numel() < numel:
            bufs[name] = None
            t = bufs[name] = self.backend.empty(numel, dtype=dtype)
        return t

    def _fused_pass(self, paths_out=None):
        be, f, eng = self.backend, self._fused, self._main_engine
        n = eng.num_paths
        plan = self.main_plan
        if plan == "auto":
            plan = "fused" if (hasattr(be, "fused_is_straight_line") and be.fused_is_straight_line(f)) else "semi"
        semi = plan == "semi" and hasattr(be, "fused_eval_paths")
        need_expo = self._fused_needs_expo or self.materialize
        paths = (paths_out if paths_out is not None else self._buffer("paths", self.sim_plan.n_dates, self.sim_plan.n_state, n)) \
            if (self.materialize or paths_out is not None or semi) else None
        bp = self.book_plan
        expo = self._buffer("expo", bp.n_netting_sets, bp.n_expo_rows, n) if (need_expo and bp.desc.want_expo) else None
        cfs = self._buffer("cfs", bp.n_netting_sets, n) if (self.materialize and bp.desc.want_cfs) else None
        # several GPUs: the records stay on the device until the ranks' records are gathered (one collective, one copy)
        kw = dict(device_records=True) if self._shard.device_collectives else {}
        if semi:
            eng.generate_paths_native(out=paths)
            rec = be.fused_eval_paths(f, paths, cfs=cfs, expo=expo, **kw)
        else:
            rec = be.fused_run(f, eng.seed, eng.path_offset, n, paths=paths, cfs=cfs, expo=expo,
                               inject_z=eng.inject_z, inject_u=eng.inject_u, **kw)
        self.last_state.update(paths=paths, cfs=cfs, expo=expo)
        return self._finish_fused_records(rec, cfs, expo, paths)

    # ---- pipelined passes ----------------------------------------------------------------------------------------------------
    # A pass of the one-launch plan ends in a few hundred bytes of accumulator records.  Gathering them over RCCL, copying them
    # to the host and merging them in Python costs tens of microseconds of latency that a 1.2 ms kernel should not wait for:
    # fused_pass_begin() enqueues the kernel on the compute stream and the gather + copy (to pinned memory) on a side stream,
    # fused_pass_end() waits for that copy only and merges.  With two passes in flight the compute stream never idles.
    def pipelined_passes_available(self) -> bool:
        be, f = self.backend, self._fused
        if f is None or self.materialize or self._fused_needs_expo or not hasattr(be, "fused_is_straight_line"):
            return False
        if self.main_plan not in ("auto", "fused") or not be.fused_is_straight_line(f):
            return False
        return self._shard.device_collectives or self._shard.world == 1

    def fused_pass_begin(self):
        be, f, eng, sh = self.backend, self._fused, self._main_engine, self._shard
        pipe = self.__dict__.setdefault("_pipe", dict(slot=0, busy=[False, False], side=torch.cuda.Stream(device=be.device), host={}))
        slot = pipe["slot"]
        if pipe["busy"][slot]:
            raise RuntimeError("fused_pass_begin: two passes are already in flight; call fused_pass_end first")
        pipe["slot"] = 1 - slot
        pipe["busy"][slot] = True
        n_rec = f.plan.n_records
        host = pipe["host"].get(slot)
        if host is None or tuple(host.shape) != (sh.world, n_rec, 4):
            host = pipe["host"][slot] = torch.empty((sh.world, n_rec, 4), dtype=torch.float64).pin_memory()
        run = lambda out: be.fused_run(f, eng.seed, eng.path_offset, eng.num_paths, inject_z=eng.inject_z, inject_u=eng.inject_u,
                                       device_records=True, records_out=out)
        if not sh.device_collectives:
            # one GPU, no process group: the merge kernel writes its few hundred bytes straight into pinned host memory (the
            # same address on the device); nothing but an event follows the kernel on the compute stream
            run(host[0])
            done = torch.cuda.Event()
            done.record()
            return (slot, host, done)
        rec = self._buffer(f"pipe_rec{slot}", n_rec, 4)
        run(rec)
        launched = torch.cuda.Event()
        launched.record()
        with torch.cuda.stream(pipe["side"]):
            pipe["side"].wait_event(launched)
            g = self._buffer(f"pipe_gather{slot}", sh.world, n_rec, 4)
            sh.all_gather_into(g, rec)                                             # RCCL orders itself after the side stream
            host.copy_(g, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        return (slot, host, done)

    def fused_pass_end(self, ticket):
        slot, host, done = ticket
        done.synchronize()
        g = host.numpy().copy()
        self._pipe["busy"][slot] = False
        self.last_state.update(paths=None, cfs=None, expo=None)
        return self._finish_fused_records(g)

    def _finish_fused_records(self, rec, cfs=None, expo=None, paths=None):
        f = self._fused
        if isinstance(rec, np.ndarray) and rec.ndim == 3:
            g = rec                                                               # already gathered: [world][n_rec][4]
        elif isinstance(rec, torch.Tensor):
            g = self._shard.all_gather_dev(rec).cpu().numpy()                     # [world][n_rec][4]
        else:
            g = self._shard.all_gather_np(rec.view(np.float64).reshape(-1, 4))
        fused_records = []
        for ns_i, lay in enumerate(f.plan.layout):
            nd = lay["n_dates"]
            fr = dict(pv=None, prof=None, cva=None, cva_metric=self._fused_cva_metric[ns_i])
            if lay["pv"] is not None:
                fr["pv"] = g[:, lay["pv"]]
            if lay["prof"] is not None:
                fr["prof"] = g[:, lay["prof"]:lay["prof"] + 2 * nd].reshape(g.shape[0], nd, 2, 4)
            if lay["cva"] is not None:
                fr["cva"] = g[:, lay["cva"]]
            fused_records.append(fr)
        return self._evaluate_all(self._shard, cfs, expo, paths, fused_records)

    def main_pass(self, paths_out=None):
        """ONE pass of the hot path over this rank's shard of the main-simulation paths: K1 path generation, K2 book
        evaluation, K4/K5 reductions (+ the accumulator collectives). Returns the nested metric results."""
        be = self.backend
        if self._fused is not None:
            return self._fused_pass(paths_out)
        paths = self._main_engine.generate_paths_native(out=paths_out)
        cfs, expo = be.eval_book(self.book, paths) if self._mc_products else (None, None)
        self.last_state.update(paths=paths, cfs=cfs, expo=expo)
        return self._evaluate_all(self._shard, cfs, expo, paths)

    @single_threaded_host()
    def run_simulation(self) -> SimulationResults:
        if self.differentiate:
            from ..aad import (_NoTangentForm, analytic_controller, run_analytic_with_autograd, run_with_bumps,
                               run_with_tangent_book, run_with_tangents, tangent_kernels_apply)
            if analytic_controller(self):
                return run_analytic_with_autograd(self)             # PVMetric(ANALYTICAL): autograd on the closed form
            if self.requires_higher_order_derivatives:
                from ..aad import run_second_order
                return run_second_order(self)                       # Monte-Carlo metrics: differences of the first-order pass
            if tangent_kernels_apply(self):
                return run_with_tangents(self)
            if self.forward_mode and hasattr(self.backend, "tangent_paths"):
                try:
                    return run_with_tangent_book(self)          # forward mode through paths, regression, book and CVA
                except _NoTangentForm:
                    pass
                except RuntimeError as e:                       # MCX_E_NOT_FUSABLE from the library: no tangent form
                    if "(-10)" not in str(e):
                        raise
            return run_with_bumps(self)
        t0 = time.perf_counter()
        be = self.backend
        self.prepare()
        t1 = time.perf_counter()
        if self._fused is not None:
            results = self._fused_pass()
            be.synchronize()
            t4 = time.perf_counter()
            self.timings = dict(preprocessing=t1 - t0, path_generation=t4 - t1, request_resolution=0.0, valuation=0.0,
                                metrics=0.0, total=t4 - t0, fused=True)
            logger.info("Simulation completed for %d netting set(s) and %d product(s): preprocessing=%.6fs "
                        "fused_main_pass=%.6fs total=%.6fs", len(self.netting_sets), len(self.products), t1 - t0, t4 - t1, t4 - t0)
            return self._package(results, [], [])
        # a book valued entirely by closed forms (PVMetric(ANALYTICAL)) needs no paths at all
        need_paths = bool(self._mc_products) or not all(m._native for m in self.risk_metrics.metrics)
        paths = self._main_engine.generate_paths_native() if need_paths else None
        be.synchronize()
        t2 = time.perf_counter()
        cfs, expo = be.eval_book(self.book, paths) if self._mc_products else (None, None)
        be.synchronize()
        t3 = time.perf_counter()
        self.last_state.update(paths=paths, cfs=cfs, expo=expo)
        results = self._evaluate_all(self._shard, cfs, expo, paths)
        t4 = time.perf_counter()
        self.timings = dict(preprocessing=t1 - t0, path_generation=t2 - t1, request_resolution=0.0,
                            valuation=t3 - t2, metrics=t4 - t3, total=t4 - t0)
        logger.info("Simulation completed for %d netting set(s) and %d product(s): preprocessing=%.6fs "
                    "path_generation=%.6fs request_resolution=%.6fs valuation=%.6fs total=%.6fs",
                    len(self.netting_sets), len(self.products), t1 - t0, t2 - t1, 0.0, t4 - t2, t4 - t0)
        return self._package(results, [], [])

    def _compile_key(self):
        """everything the compiled descriptors depend on besides the (immutable) object graph of this controller: the model
        parameter VALUES (bumped runs change them), the smoothing flag and the backend"""
        return (tuple(float(p.detach()) for p in self.model.get_model_params()), bool(self.model.perform_smoothing), id(self.backend),
                bool(self.materialize), bool(self.reference_float32_cf_cache), repr(getattr(self.regression_function, "degree", None)),
                len(self.products), tuple(type(m).__name__ for m in self.risk_metrics.metrics))

    def invalidate(self):
        """forget the compiled descriptors / uploaded book (after mutating a product, a metric or a netting set)"""
        self._compiled_key = None
        self._pred_memo = {}

    def _compile_all(self):
        """objects -> descriptors, LSM atoms registered before the plan is frozen and uploaded.  A controller that is run again
        with unchanged model parameters reuses the descriptors and the uploaded book of its previous run (the host-side
        compilation of a 120-date Bermudan swaption is ~25 ms of Python against ~35 ms of GPU work)."""
        key = self._compile_key()
        if self.reuse_compiled and getattr(self, "_compiled_key", None) == key and getattr(self, "book", None) is not None:
            self.backend.book_reset_coeffs(self.book, self._coeffs_at_upload)
            return
        self._compile()
        if self.requires_regression:
            self._register_regression_atoms()
        self.book_plan = BookPlan(self._comp, *self._plan_args)
        if len(self.book_plan.events) >= (1 << 16):
            # a big book's flattening + upload (mcx_book_create: ~35 ms for 4 x 10^5 events, in C, the GIL released) runs beside the
            # host work that does not need the book yet — the sub-step plan, the pre-simulation launch, the regression job lists;
            # _book_ready() joins before the first call that takes the book
            box: dict = {}

            def work(plan=self.book_plan, be=self.backend):
                try:
                    box["book"] = be.book_create(plan)
                except BaseException as e:                                # noqa: BLE001 — re-raised by _book_ready
                    box["err"] = e
            th = threading.Thread(target=work, name="mcx-book-create")
            th.start()
            self.book, self._book_pending = None, (th, box)
        else:
            self.book = self.backend.book_create(self.book_plan)
        self._coeffs_at_upload = self.book_plan.coeffs.copy()
        self._compiled_key = key

    def _book_ready(self):
        """the uploaded book (waits for a mcx_book_create still running beside the planning, see _compile_all)"""
        pend = self.__dict__.pop("_book_pending", None)
        if pend is not None:
            th, box = pend
            th.join()
            if "err" in box:
                self._compiled_key = None
                raise box["err"]
            self.book = box["book"]
        return self.book

    def _evaluate_all(self, shard, cfs, expo, paths, fused_records=None):
        analytical = [[0.0 for _ in self.risk_metrics.metrics] for _ in self.netting_sets]
        has_pathwise = [False] * len(self.netting_sets)
        for p_i, p in enumerate(self.products):
            ns_i = self.product_to_netting_set_idx[p_i]
            if self._can_skip_monte_carlo_for_product(p):
                for m_i, m in enumerate(self.risk_metrics.metrics):
                    analytical[ns_i][m_i] += float(m.evaluate_analytically(product=p, model=self.model)[0][0])
            else:
                has_pathwise[ns_i] = True
        return [self._evaluate_netting_set(shard, i, ns, cfs, expo, paths, has_pathwise[i], analytical[i],
                                           fused_records[i] if fused_records else None)
                for i, ns in enumerate(self.netting_sets)]

    def _package(self, results, grads, higher):
        return SimulationResults(
            results, grads, higher,
            netting_set_names=self._make_unique_names([ns.get_name() for ns in self.netting_sets]),
            metric_names=self._make_unique_names([m.get_name() for m in self.risk_metrics.metrics]),
            model_param_names=self.model.get_model_param_names())
