"""CPU-oracle backend with the same method surface as mcx._native.HipBackend.

TEST INFRASTRUCTURE: lets the `-m "not gpu"` suite run the product's HOST logic (descriptor compilation, LSM solve,
metric finalisation, radix-select driver, sharding) against oracle/liborc.so and pin both to the reference's golden
vectors.  The product never selects this backend."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
ORC_LIB = os.path.join(ORC_DIR, "liborc.so")


def build_oracle():
    if os.environ.get("MCX_ORACLE_LIB"):                 # the sanitizer build of the checker (oracle/Makefile, SAN=1)
        return os.path.abspath(os.environ["MCX_ORACLE_LIB"])
    src = os.path.join(ORC_DIR, "mcx_oracle.c")
    if not os.path.exists(ORC_LIB) or os.path.getmtime(ORC_LIB) < max(os.path.getmtime(src),
                                                                      os.path.getmtime(os.path.join(ROOT, "include", "mcx.h"))):
        subprocess.check_call(["make", "-s", "-C", ORC_DIR], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return ORC_LIB


def load_oracle() -> C.CDLL:
    lib = C.CDLL(build_oracle())
    lib.orc_philox4x32_10.restype = None
    lib.orc_draw.restype = None
    lib.orc_set_bridge_rng.restype = None
    return lib


def _p(t) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, torch.Tensor):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


class _Obj:
    def __init__(self, plan):
        self.plan, self.ptr = plan, None


class OracleBackend:
    name = "oracle"

    def __init__(self):
        import mcx._abi as _abi
        self._abi = _abi
        self.lib = load_oracle()
        self.device = torch.device("cpu")

    def empty(self, *shape, dtype=torch.float64):
        return torch.empty(*shape, dtype=dtype)

    def zeros(self, *shape, dtype=torch.float64):
        return torch.zeros(*shape, dtype=dtype)

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def synchronize(self):
        pass

    def sim_create(self, plan):
        return _Obj(plan)

    def generate_paths(self, sim, seed, path_offset, n_paths, inject_z=None, inject_u=None, out=None, init_state=None):
        plan = sim.plan
        if out is None:
            out = torch.empty(plan.n_dates, plan.n_state, n_paths, dtype=torch.float64)
        if init_state is not None:
            rc = self.lib.orc_generate_paths_from_state(C.byref(plan.desc), C.c_uint64(seed), C.c_uint64(path_offset), C.c_int64(n_paths),
                                                        C.c_int64(n_paths), _p(init_state), _p(out), _p(inject_z), _p(inject_u))
            assert rc == 0
            return out
        rc = self.lib.orc_generate_paths(C.byref(plan.desc), C.c_uint64(seed), C.c_uint64(path_offset), C.c_int64(n_paths),
                                         C.c_int64(n_paths), _p(out), _p(inject_z), _p(inject_u))
        assert rc == 0
        return out

    def book_create(self, plan):
        return _Obj(plan)

    def book_set_exercise_replay(self, book, mode, bits):
        self.lib.orc_set_exercise_replay(C.c_int(mode), _p(bits) if mode else None, C.c_int64(bits.shape[1] if mode else 0))
        self._keep_replay = bits

    def new_exercise_bits(self, n_events, n_paths):
        return torch.zeros((n_events, n_paths), dtype=torch.uint8)

    def book_reset_coeffs(self, book, values):
        self.book_set_coeffs(book, 0, values)

    def book_set_coeffs(self, book, offset, values):
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        book.plan.coeffs[offset:offset + v.size] = v

    def book_set_bridge_rng(self, book, seed, path_offset, inject=None):
        table, ld = None, 0
        if inject:
            arr = (C.c_void_p * book.plan.desc.n_products)()
            for p_i, t in inject.items():
                arr[p_i] = t.data_ptr()
                ld = t.shape[1]
            table = arr
            self._keep_bridge = (arr, dict(inject))
        self.lib.orc_set_bridge_rng(C.c_uint64(seed), C.c_uint64(path_offset), table, C.c_int64(ld))

    def eval_book(self, book, paths):
        plan = book.plan
        n = paths.shape[2]
        cfs = torch.empty(plan.n_netting_sets, n, dtype=torch.float64) if plan.desc.want_cfs else None
        expo = torch.empty(plan.n_netting_sets, plan.n_expo_rows, n, dtype=torch.float64) if plan.desc.want_expo else None
        rc = self.lib.orc_eval_book(C.byref(plan.desc), C.c_int64(plan.n_state), _p(paths), C.c_int64(n), C.c_int64(n),
                                    _p(cfs), _p(expo), C.c_int64(n))
        assert rc == 0
        return cfs, expo

    def resolve_atoms(self, book, atom_ids, paths):
        ids = np.ascontiguousarray(atom_ids, dtype=np.int32)
        n = paths.shape[2]
        out = torch.empty(len(ids), n, dtype=torch.float64)
        self.lib.orc_resolve_atoms(C.byref(book.plan.desc), C.c_int64(book.plan.n_state), _p(ids), C.c_int32(len(ids)),
                                   _p(paths), C.c_int64(n), C.c_int64(n), _p(out), C.c_int64(n))
        return out

    def lsm_stats(self, book, atom_ids, paths):
        ids = np.ascontiguousarray(atom_ids, dtype=np.int32)
        n = paths.shape[2]
        out = np.zeros((len(ids), 2))
        self.lib.orc_lsm_stats(C.byref(book.plan.desc), C.c_int64(book.plan.n_state), _p(ids), C.c_int32(len(ids)),
                               _p(paths), C.c_int64(n), C.c_int64(n), _p(out))
        return out

    def lsm_step(self, book, product, roll_begin, roll_end, num_atom, x_atom, shift, scale, paths, W, flags=0):
        n = paths.shape[2]
        K, S = book.plan.n_basis, W.shape[0]
        mom = torch.empty((2 * K - 1) + S * K, dtype=torch.float64)
        self.lib.orc_lsm_step(C.byref(book.plan.desc), C.c_int64(book.plan.n_state), C.c_int32(product),
                              C.c_int32(roll_begin), C.c_int32(roll_end), C.c_int32(num_atom), C.c_int32(x_atom),
                              C.c_double(shift), C.c_double(scale), _p(paths), C.c_int64(n), C.c_int64(n), _p(W),
                              C.c_int64(W.shape[1]), _p(mom), C.c_int32(flags))
        return mom

    def lsm_step_batch(self, book, jobs, n_states, paths, W, ld_w, flags=0):
        """checker composition of orc_lsm_step over the jobs of one batch (the product's mcx_lsm_step_batch is one launch)"""
        n = paths.shape[2]
        K, S = book.plan.n_basis, n_states
        out = np.zeros((len(jobs), (2 * K - 1) + S * K))
        for j, q in enumerate(jobs):
            Wp = W[int(q["w_offset"]):int(q["w_offset"]) + S * ld_w].view(S, ld_w)
            out[j] = self.lsm_step(book, int(q["product"]), int(q["roll_begin"]), int(q["roll_end"]), int(q["num_atom"]),
                                   int(q["x_atom"]), float(q["shift"]), float(q["scale"]), paths, Wp, flags).numpy()
        return out

    def book_set_coeffs_batch(self, book, offsets, values):
        for off, v in zip(offsets, values):
            self.book_set_coeffs(book, int(off), v)

    def reduce_vector(self, x):
        out = np.zeros(1, dtype=self._abi.ACC_DTYPE)
        self.lib.orc_reduce_vector(_p(x), C.c_int64(x.shape[0]), _p(out))
        return out

    def reduce_profiles(self, unsec, expo_ns):
        n = expo_ns.shape[1]
        out = np.zeros((unsec.n_dates, 2), dtype=self._abi.ACC_DTYPE)
        self.lib.orc_reduce_profiles(C.byref(unsec.desc), _p(expo_ns), C.c_int64(n), C.c_int64(n), _p(out))
        return out

    def reduce_cva(self, book, unsec, surv_atoms, cond_atoms, recovery, expo_ns, paths):
        n = expo_ns.shape[1]
        sa = np.ascontiguousarray(surv_atoms, dtype=np.int32)
        ca = np.ascontiguousarray(cond_atoms, dtype=np.int32)
        out = np.zeros(1, dtype=self._abi.ACC_DTYPE)
        self.lib.orc_reduce_cva(C.byref(book.plan.desc), C.c_int64(book.plan.n_state), C.byref(unsec.desc), _p(sa), _p(ca),
                                C.c_double(recovery), _p(expo_ns), _p(paths), C.c_int64(n), C.c_int64(n),
                                C.c_int64(paths.shape[2]), _p(out))
        return out

    def unsecured(self, unsec, expo_ns):
        n = expo_ns.shape[1]
        out = torch.empty(unsec.n_dates, n, dtype=torch.float64)
        self.lib.orc_unsecured(C.byref(unsec.desc), _p(expo_ns), C.c_int64(n), C.c_int64(n), _p(out), C.c_int64(n))
        return out

    def select_hist(self, unsec, expo_ns, n_sel, prefix, shift, bits):
        n = expo_ns.shape[1]
        pf = np.ascontiguousarray(prefix, dtype=np.uint64)
        hist = torch.empty(unsec.n_dates, n_sel, 1 << bits, dtype=torch.int64)
        self.lib.orc_select_hist(C.byref(unsec.desc), _p(expo_ns), C.c_int64(n), C.c_int64(n), C.c_int32(n_sel), _p(pf),
                                 C.c_int32(shift), C.c_int32(bits), _p(hist))
        return hist

    def tangent_european(self, sim, opts, n_ns, n_params, seed, path_offset, n_paths, inject_z=None, inject_u=None):
        cfs = torch.empty(n_ns, n_paths, dtype=torch.float64)
        dcfs = torch.empty(n_ns, n_params, n_paths, dtype=torch.float64)
        rc = self.lib.orc_tangent_european(C.byref(sim.plan.desc), opts, C.c_int32(len(opts)), C.c_int32(n_ns),
                                           C.c_uint64(seed), C.c_uint64(path_offset), C.c_int64(n_paths), C.c_int64(n_paths),
                                           _p(cfs), _p(dcfs), C.c_int64(n_paths), _p(inject_z), _p(inject_u))
        assert rc == 0
        return cfs, dcfs

    # the "fused pass" of the oracle is simply the composition of its primitives with the same record layout
    def fused_create(self, sim, book, plan):
        f = _Obj(plan)
        f.sim, f.book = sim, book
        return f

    def fused_run(self, fused, seed, path_offset, n_paths, paths=None, cfs=None, expo=None, inject_z=None, inject_u=None):
        p = self.generate_paths(fused.sim, seed, path_offset, n_paths, inject_z, inject_u, out=paths)
        return self.fused_eval_paths(fused, p, cfs=cfs, expo=expo)

    def fused_eval_paths(self, fused, paths, cfs=None, expo=None):
        from mcx.plan import UnsecuredSpec
        plan = fused.plan
        p = paths
        c, e = self.eval_book(fused.book, p)
        if cfs is not None and c is not None:
            cfs.copy_(c)
        if expo is not None and e is not None:
            expo.copy_(e)
        out = np.zeros(plan.n_records, dtype=self._abi.ACC_DTYPE)
        for k, (lay, sp) in enumerate(zip(plan.layout, plan.specs)):
            unsec = UnsecuredSpec(sp["rows"], None, sp["threshold"], False)
            nd = len(sp["rows"])
            if lay["pv"] is not None:
                out[lay["pv"]] = self.reduce_vector(c[k])[0]
            if lay["prof"] is not None:
                out[lay["prof"]:lay["prof"] + 2 * nd] = self.reduce_profiles(unsec, e[k]).reshape(-1)
            if lay["cva"] is not None:
                out[lay["cva"]] = self.reduce_cva(fused.book, unsec, sp["surv"], sp["cond"], sp["recovery"], e[k], p)[0]
        return out

    # straight sort-based restatement of pfe_metric.py:49-73 (checks the radix select)
    def pfe_sort(self, unsec, expo_ns, q_index):
        n = expo_ns.shape[1]
        out = np.zeros((unsec.n_dates, 3))
        self.lib.orc_pfe_sort(C.byref(unsec.desc), _p(expo_ns), C.c_int64(n), C.c_int64(n), C.c_int64(q_index), _p(out))
        return out
