"""Host-side compilation of the reference's object graph into the POD descriptors of include/mcx.h.

  SimPlan   <- (model, simulation timeline, scheme, num_steps)      drives K1 (engine/engine.py:35-123 in the reference)
  BookPlan  <- (products, netting sets, exposure grid, metrics)     drives K2/K3/K4 (controller.py:294-471)

Everything here is tiny host work (float-keyed timelines, closed-form coefficients); it must reproduce the reference's
float arithmetic exactly because later lookups are keyed on these floats (SURVEY.md "float-keyed timelines").
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _abi
from .common.enums import SimulationScheme


class SimPlan:
    """Sub-step table of MonteCarloEngine's time loop.

    Reference semantics kept (engine/engine.py:46-62): t_prev starts at the calibration date and advances by repeated
    `+ dt` (never snapped to the timeline date); a timeline date with dt <= 0 (the calibration date itself) is stored
    without stepping; models see delta_t = (t_prev + dt) - t_prev while the ANALYTICAL Cholesky factor is keyed on dt."""

    def __init__(self, model, timeline, scheme: SimulationScheme, num_steps: int):
        if not model._supports_scheme(scheme):
            raise NotImplementedError(f"{type(model).__name__} does not implement the {scheme.name} scheme")
        self.model, self.scheme, self.num_steps = model, scheme, int(num_steps)
        self.timeline = np.asarray(timeline, dtype=np.float64).reshape(-1)
        slots = model._slots()
        if len(slots) > _abi.MAX_SLOTS:
            raise ValueError(f"at most {_abi.MAX_SLOTS} sub-models are supported")
        self.n_slots = len(slots)
        self.n_state = sum(s.state_dim for s in slots)
        self.n_z = sum(s.sim_dim for s in slots)
        if self.n_z > _abi.MAX_Z or self.n_state > _abi.MAX_STATE:
            raise ValueError("simulation / state dimension exceeds MCX_MAX_Z / MCX_MAX_STATE")
        self.n_uniform = int(getattr(model, "_n_uniform", lambda s: 0)(scheme))
        self.n_dates = len(self.timeline)

        steps, aux, chols, chol_key = [], [], [], {}
        t_prev = np.float64(model.t0())
        self.n_initial_store = 0
        for ti, t_now in enumerate(self.timeline):
            dt = (t_now - t_prev) / np.float64(self.num_steps)
            if dt > 0:
                for _ in range(self.num_steps):
                    t2 = t_prev + dt
                    dt_model = float(t2 - t_prev)
                    key = float(dt) if scheme == SimulationScheme.ANALYTICAL else None
                    if key not in chol_key:
                        chol_key[key] = len(chols)
                        chols.append(model.get_cholesky(scheme, float(dt)).detach().cpu().numpy().astype(np.float64))
                    steps.append((dt_model, math.sqrt(dt_model), float(t_prev), -1, chol_key[key]))
                    per_slot = model._step_aux(scheme, float(t_prev), dt_model)
                    row = np.zeros((self.n_slots, _abi.AUX))
                    for s, vals in enumerate(per_slot):
                        row[s, :len(vals)] = vals
                    aux.append(row)
                    t_prev = t2
                steps[-1] = steps[-1][:3] + (ti,) + steps[-1][4:]
            else:
                if steps:
                    raise ValueError("non-increasing simulation timeline")
                self.n_initial_store += 1
        self.steps = np.array(steps, dtype=_abi.STEP_DTYPE) if steps else np.zeros(0, dtype=_abi.STEP_DTYPE)
        self.n_steps = len(steps)
        self.aux = np.ascontiguousarray(np.stack(aux)) if aux else np.zeros((0, self.n_slots, _abi.AUX))
        self.chol = np.ascontiguousarray(np.stack(chols)) if chols else np.zeros((0, self.n_z, self.n_z))
        self.init_state = np.array(model._initial_state(), dtype=np.float64)
        self.flags = _abi.FLAG_SMOOTHING if model.perform_smoothing else 0

        d = _abi.SimDesc()
        d.scheme, d.n_slots, d.n_state, d.n_z = scheme.value, self.n_slots, self.n_state, self.n_z
        d.n_uniform, d.n_steps, d.n_dates, d.n_chol = self.n_uniform, self.n_steps, self.n_dates, len(chols)
        d.n_initial_store, d.flags = self.n_initial_store, self.flags
        so = zo = 0
        for i, s in enumerate(slots):
            d.slots[i].kind, d.slots[i].state_off, d.slots[i].z_off, d.slots[i].flags = s.kind, so, zo, s.flags
            for j, v in enumerate(s.params):
                d.slots[i].p[j] = v
            so += s.state_dim
            zo += s.sim_dim
        d.steps, d.chol, d.aux, d.init_state = (_abi.ptr(self.steps), _abi.ptr(self.chol), _abi.ptr(self.aux),
                                                _abi.ptr(self.init_state))
        self.desc = d                      # keeps pointers into the numpy arrays above (kept alive by self)

    @property
    def path_steps_per_path(self) -> int:
        return self.n_steps

    @classmethod
    def single_step(cls, model, scheme: SimulationScheme, t1: float, t2: float) -> "SimPlan":
        """the one sub-step [t1, t2] with an identity Cholesky factor: Model.simulate_time_step_*(time1, time2, state, corr_randn)
        receives normals that are already correlated (model.py:38-48 applies the factor before the step)"""
        plan = cls(model, np.array([t2]), scheme, 1)
        if plan.n_steps != 1:
            raise ValueError("time2 must lie after the calibration date")
        dt = float(t2) - float(t1)
        plan.steps["dt"], plan.steps["sqrt_dt"], plan.steps["t1"] = dt, math.sqrt(dt), float(t1)
        plan.aux[:] = 0.0
        for s, vals in enumerate(model._step_aux(scheme, float(t1), dt)):
            plan.aux[0, s, :len(vals)] = vals
        plan.chol[:] = np.eye(plan.n_z)[None]
        return plan


_EV_WORDS = _abi.EVENT_DTYPE.itemsize // 8


class BookCompiler:
    """Accumulates atoms / terms / events while products and metrics describe themselves."""

    def __init__(self, model, sim_timeline, n_basis: int):
        self.model = model
        self.time_to_index = {float(t): i for i, t in enumerate(sim_timeline)}
        self.n_basis = n_basis
        self.atoms: list[tuple] = []
        self.atom_src: list = []                    # (request, asset_id) behind every atom (None: constant) — re-evaluated under bumped models
        self._atom_key: dict = {}
        self.terms: list[tuple] = []
        self.n_events = 0
        self._ev_cap = 4096
        self._ev = np.empty(self._ev_cap, dtype=_abi.EVENT_DTYPE)
        self._ev_raw = self._ev.view(np.float64).reshape(-1, _EV_WORDS)
        self.coeff_init: dict[int, float] = {}      # constants parked in the coefficient array (bridge-barrier parameters)

    def tidx(self, time) -> int:
        return self.time_to_index[float(time)]

    def atom(self, req, asset_id, time) -> int:
        ti = self.time_to_index[float(time)]
        key = (ti, asset_id, req.request_type.value, req.id, req.time1, req.time2)
        hit = self._atom_key.get(key)
        if hit is None:
            co = self.model._atom(req, asset_id)
            hit = self._atom_key[key] = len(self.atoms)
            self.atoms.append((ti, -1 if co.col is None else co.col, co.a, co.d, co.b, co.c0, co.c1))
            self.atom_src.append((req, asset_id))
        return hit

    def const_atom(self, value: float) -> int:
        key = ("const", float(value))
        if key not in self._atom_key:
            self._atom_key[key] = len(self.atoms)
            self.atoms.append((0, -1, float(value), 0.0, 0.0, 0.0, 0.0))
            self.atom_src.append(None)
        return self._atom_key[key]

    def add_terms(self, terms) -> tuple[int, int]:
        """terms of one event; weights of identical (atom, denominator) pairs are merged first — the reference's leg-by-leg
        sums revisit the same zero-bond price (e.g. the float leg's N*(P_j - P_{j+1}) telescopes), and every distinct atom
        costs one exp per path on the GPU.  Only the summation order changes (<= 1e-16 relative)."""
        merged: dict[tuple[int, int], float] = {}
        for t in terms:
            key = (int(t[1]), int(t[2]) if len(t) > 2 else -1)
            merged[key] = merged.get(key, 0.0) + float(t[0])
        b = len(self.terms)
        for (atom, den), w in merged.items():
            if w != 0.0:
                self.terms.append((w, atom, den))
        return b, len(self.terms)

    # events are written straight into one growing EVENT_DTYPE array (doubling): a 5,000-product book has 4 x 10^5 of them, as
    # ~10^4 single rows interleaved with ~10^4 array blocks — collecting them in a Python list and converting run by run was a
    # quarter of the compile time
    def reserve_events(self, n: int) -> None:
        """room for n events up front (the controller knows the count: growing by doubling touches every page several times)"""
        if n > self._ev_cap:
            self._ev_grow((n + 1) // 2)

    def _ev_grow(self, need: int) -> None:
        new = np.empty(2 * need, dtype=_abi.EVENT_DTYPE)
        raw = new.view(np.float64).reshape(-1, _EV_WORDS)
        raw[:self.n_events] = self._ev_raw[:self.n_events]
        self._ev, self._ev_raw, self._ev_cap = new, raw, len(new)

    def add_event(self, kind, t_idx, num_atom, x_atom, term_range, coeff_off, expo_row, strike=0.0, sign=1.0,
                  aux=(0.0, 0.0, 0.0, 0.0)) -> int:
        return self.add_event_row((kind, t_idx, num_atom, x_atom, term_range[0], term_range[1], coeff_off, expo_row,
                                   float(strike), float(sign), tuple(aux)))

    def add_event_row(self, row: tuple) -> int:
        """an event as the ready EVENT_DTYPE tuple (a cash event is listed twice per product: built once, added twice)"""
        if self.n_events >= self._ev_cap:
            self._ev_grow(self.n_events + 1)
        self._ev[self.n_events] = row
        self.n_events += 1
        return self.n_events - 1

    def add_event_block(self, block: np.ndarray) -> None:
        """a run of events as one array — EVENT_DTYPE records, or their raw [n][10] float64 image (the per-(product, exposure date)
        events of big books: 10^6 of them).  Copied as 8-byte words: numpy assigns structured records field by field, 30x slower."""
        n = len(block)
        if n:
            if self.n_events + n > self._ev_cap:
                self._ev_grow(self.n_events + n)
            if block.dtype != np.float64:
                block = block.view(np.float64).reshape(-1, _EV_WORDS)
            self._ev_raw[self.n_events:self.n_events + n] = block
            self.n_events += n

    def events_array(self) -> np.ndarray:
        if self.n_events == 0:
            return np.zeros(0, dtype=_abi.EVENT_DTYPE)
        return np.ascontiguousarray(self._ev[:self.n_events])


class BookPlan:
    """Frozen descriptor arrays + bookkeeping the controller needs (coefficient offsets, atom ids of requests)."""

    def __init__(self, comp: BookCompiler, products: np.ndarray, n_netting_sets: int, n_expo_rows: int, n_coeffs: int,
                 want_cfs: bool, want_expo: bool, n_state: int):
        self.atoms = np.array(comp.atoms, dtype=_abi.ATOM_DTYPE) if comp.atoms else np.zeros(0, dtype=_abi.ATOM_DTYPE)
        self.terms = np.array(comp.terms, dtype=_abi.TERM_DTYPE) if comp.terms else np.zeros(0, dtype=_abi.TERM_DTYPE)
        self.events = comp.events_array()
        self.products = products
        self.coeffs = np.zeros(max(n_coeffs, 1), dtype=np.float64)
        for off, val in comp.coeff_init.items():
            self.coeffs[off] = val
        self.n_netting_sets, self.n_expo_rows, self.n_basis = n_netting_sets, n_expo_rows, comp.n_basis
        self.n_state = n_state
        d = _abi.BookDesc()
        d.n_atoms, d.n_terms, d.n_events, d.n_products = len(self.atoms), len(self.terms), len(self.events), len(products)
        d.n_netting_sets, d.n_expo_rows, d.n_basis, d.n_coeffs = n_netting_sets, n_expo_rows, comp.n_basis, len(self.coeffs)
        d.want_cfs, d.want_expo = int(want_cfs), int(want_expo)
        d.n_state = n_state
        d.n_dates = len(comp.time_to_index)
        d.atoms, d.terms, d.events, d.products = (_abi.ptr(self.atoms), _abi.ptr(self.terms), _abi.ptr(self.events),
                                                  _abi.ptr(self.products))
        d.coeffs = _abi.ptr(self.coeffs)
        self.desc = d


class UnsecuredSpec:
    """mcx_unsecured_desc of one netting set (products/netting_set.py:156-184)."""

    def __init__(self, rows, delayed, threshold: float, collateralized: bool):
        self.rows = np.ascontiguousarray(rows, dtype=np.int32)
        self.delayed = None if delayed is None else np.ascontiguousarray(delayed, dtype=np.int32)
        self.n_dates = len(self.rows)
        d = _abi.UnsecuredDesc()
        d.n_dates, d.collateralized, d.threshold = self.n_dates, int(collateralized), float(threshold)
        d.row, d.delayed = _abi.ptr(self.rows), _abi.ptr(self.delayed)
        self.desc = d


def solve_normal_equations(moments: np.ndarray, K: int, S: int, shift: float, scale: float, degenerate: bool,
                           x0: float) -> np.ndarray:
    """Least-squares coefficients [S][K] in the RAW monomial basis from the moments of the shifted/scaled basis
    z = (x - shift) * scale  (what torch.linalg.lstsq(A, Y) returns in the reference, controller.py:370-374).

    degenerate (all paths share x = x0, e.g. the t = calibration-date regression): the reference's gelsy driver
    returns the minimum-norm solution of the rank-1 system, i.e. c = v * mean(Y) / (v.v) with v = [1, x0, .., x0^(K-1)].
    """
    m = np.asarray(moments, dtype=np.float64)
    n = m[0]
    out = np.zeros((S, K))
    if n <= 0:
        return out
    if degenerate:
        v = np.array([x0 ** k for k in range(K)])
        for s in range(S):
            mean_y = m[(2 * K - 1) + s * K] / n
            out[s] = v * (mean_y / float(v @ v))
        return out
    G = np.array([[m[j + k] for k in range(K)] for j in range(K)])
    rhs = np.array([[m[(2 * K - 1) + s * K + k] for k in range(K)] for s in range(S)]).T     # [K][S]
    try:
        b = np.linalg.solve(G, rhs)
    except np.linalg.LinAlgError:
        b = np.linalg.lstsq(G, rhs, rcond=None)[0]
    # z^k = scale^k (x - shift)^k  -> expand into monomials of x
    T = np.zeros((K, K))            # T[j][k] = coefficient of x^j in z^k
    for k in range(K):
        for j in range(k + 1):
            T[j, k] = (scale ** k) * math.comb(k, j) * ((-shift) ** (k - j))
    return (T @ b).T


def solve_normal_equations_batch(moments: np.ndarray, K: int, S: int, shift: np.ndarray, scale: np.ndarray,
                                 degenerate: np.ndarray, x0: np.ndarray) -> np.ndarray:
    """`solve_normal_equations` for many (product, date) systems at once: moments [n][NM] -> coefficients [n][S][K]."""
    m = np.asarray(moments, dtype=np.float64)
    n_jobs = m.shape[0]
    out = np.zeros((n_jobs, S, K))
    if n_jobs == 0:
        return out
    idx = np.arange(K)[:, None] + np.arange(K)[None, :]
    G = m[:, idx]                                                              # [n][K][K]
    rhs = m[:, (2 * K - 1):(2 * K - 1) + S * K].reshape(n_jobs, S, K).transpose(0, 2, 1)    # [n][K][S]
    regular = (~np.asarray(degenerate, dtype=bool)) & (m[:, 0] > 0)
    if regular.any():
        r = np.nonzero(regular)[0]
        try:
            b = np.linalg.solve(G[r], rhs[r])
        except np.linalg.LinAlgError:
            b = np.stack([np.linalg.lstsq(G[q], rhs[q], rcond=None)[0] for q in r])
        T = np.zeros((len(r), K, K))
        for k in range(K):
            for j in range(k + 1):
                T[:, j, k] = (scale[r] ** k) * math.comb(k, j) * ((-shift[r]) ** (k - j))
        out[r] = np.matmul(T, b).transpose(0, 2, 1)
    for q in np.nonzero(np.asarray(degenerate, dtype=bool) & (m[:, 0] > 0))[0]:
        out[q] = solve_normal_equations(m[q], K, S, float(shift[q]), float(scale[q]), True, float(x0[q]))
    return out


class FusedPlan:
    """mcx_fused_desc: which accumulator records the fused pass produces, per netting set, and where they sit in the
    flat record array: [PV][2*n_dates profiles][CVA] per netting set, in netting-set order."""

    def __init__(self, ns_specs: list[dict], want_pv: bool, row_t_idx):
        self.row_t_idx = np.ascontiguousarray(row_t_idx, dtype=np.int32)
        self.specs = ns_specs
        self.want_pv = bool(want_pv)
        self._keep = []
        arr = (_abi.FusedNsDesc * len(ns_specs))()
        self.layout = []            # per netting set: dict(pv=idx|None, prof=idx|None, cva=idx|None, n_dates)
        pos = 0
        for k, sp in enumerate(ns_specs):
            rows = np.ascontiguousarray(sp["rows"], dtype=np.int32)
            surv = np.ascontiguousarray(sp["surv"], dtype=np.int32) if sp.get("surv") is not None else None
            cond = np.ascontiguousarray(sp["cond"], dtype=np.int32) if sp.get("cond") is not None else None
            self._keep += [rows, surv, cond]
            d = arr[k]
            d.netting_set, d.n_dates = k, len(rows)
            d.want_profiles, d.want_cva = int(sp["want_profiles"]), int(surv is not None)
            d.threshold, d.recovery = float(sp["threshold"]), float(sp.get("recovery", 0.0))
            d.row, d.surv_atoms, d.cond_atoms = _abi.ptr(rows), _abi.ptr(surv), _abi.ptr(cond)
            lay = dict(pv=None, prof=None, cva=None, n_dates=len(rows))
            if self.want_pv:
                lay["pv"] = pos
                pos += 1
            if sp["want_profiles"]:
                lay["prof"] = pos
                pos += 2 * len(rows)
            if surv is not None:
                lay["cva"] = pos
                pos += 1
            self.layout.append(lay)
        self.n_records = pos
        self._ns_array = arr
        d = _abi.FusedDesc()
        d.n_netting_sets, d.want_pv, d.n_expo_rows = len(ns_specs), int(self.want_pv), len(self.row_t_idx)
        d.row_t_idx, d.ns = _abi.ptr(self.row_t_idx), C.cast(arr, C.c_void_p)
        self.desc = d
