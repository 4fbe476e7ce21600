"""ctypes binding of libmcx_hip.so — the ONLY compute backend of the product.

There is no CPU fallback: if the HIP library is missing, or no GPU is visible, construction fails loudly.
PyTorch-ROCm is used as the container of device memory (torch.empty(device="cuda")) and for the current HIP stream;
every numeric kernel on the hot path is hand-written HIP behind the C ABI of include/mcx.h.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCX_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "csrc", "libmcx_hip.so")   # override: kernel experiments

_EXPORTS = [
    "mcx_abi_version", "mcx_create", "mcx_destroy", "mcx_last_error", "mcx_device_info",
    "mcx_sim_create", "mcx_sim_destroy", "mcx_generate_paths", "mcx_generate_paths_from_state", "mcx_rng_draws",
    "mcx_comm_unique_id", "mcx_comm_init", "mcx_comm_destroy", "mcx_allreduce_f64", "mcx_allgather_f64",
    "mcx_book_create", "mcx_book_destroy", "mcx_book_set_coeffs", "mcx_eval_book", "mcx_resolve_atoms",
    "mcx_lsm_stats", "mcx_lsm_step", "mcx_lsm_run", "mcx_lsm_solve", "mcx_lsm_step_batch", "mcx_lsm_step_batch_dev", "mcx_lsm_solve_batch", "mcx_lsm_run_batch", "mcx_book_get_coeffs", "mcx_book_set_coeffs_batch", "mcx_book_set_bridge_rng", "mcx_book_set_exercise_replay",
    "mcx_fused_is_straight_line", "mcx_tangent_paths", "mcx_tangent_lsm", "mcx_tangent_lsm_step", "mcx_tangent_eval", "mcx_tangent_cva", "mcx_tangent_profiles", "mcx_tangent_pick",
    "mcx_box_muller", "mcx_tangent_european", "mcx_fused_create", "mcx_fused_destroy", "mcx_fused_num_records", "mcx_fused_run", "mcx_fused_eval_paths", "mcx_fused_run_device", "mcx_fused_eval_paths_device", "mcx_fused_set_timing", "mcx_fused_kernel_times",
    "mcx_value_poly_fit", "mcx_book_collapse_values", "mcx_book_value_poly_info", "mcx_rows_minmax",
    "mcx_reduce_vector", "mcx_reduce_profiles", "mcx_reduce_cva", "mcx_unsecured", "mcx_select_hist", "mcx_select_hist_dev", "mcx_select_narrow", "mcx_select_bracket", "mcx_select_hist_rows",
]


def load_library(path: str = LIB_PATH) -> C.CDLL:
    if not os.path.exists(path):
        raise RuntimeError(f"HIP extension not built: {path} is missing. Run `python -c 'import __graft_entry__ as g; "
                           f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name in _EXPORTS:
        if not hasattr(lib, name):
            raise RuntimeError(f"{path} does not export {name}")
    lib.mcx_last_error.restype = C.c_char_p
    lib.mcx_last_error.argtypes = [C.c_void_p]
    lib.mcx_destroy.restype = None
    lib.mcx_sim_destroy.restype = None
    lib.mcx_book_destroy.restype = None
    lib.mcx_fused_destroy.restype = None
    lib.mcx_fused_num_records.argtypes = [C.c_void_p]
    lib.mcx_fused_is_straight_line.argtypes = [C.c_void_p]
    if lib.mcx_abi_version() != _abi.ABI_VERSION:
        raise RuntimeError("libmcx_hip.so ABI version mismatch")
    return lib


def _vp(x) -> C.c_void_p:
    return C.c_void_p(int(x))


def _ld(t: torch.Tensor) -> int:
    """row stride (doubles) of a [..][rows][paths] tensor whose rows are evenly spaced and path-contiguous (the paths tensor may
    carry a padded leading dimension: padded_ld)"""
    assert t.stride(-1) == 1 and all(t.stride(k) == t.shape[k + 1] * t.stride(k + 1) for k in range(t.dim() - 2))
    return int(t.stride(-2)) if t.dim() >= 2 else int(t.shape[-1])


def _ld_out(cfs, expo, n: int) -> int:
    """the one leading dimension the library takes for the cashflow [ns][n] and exposure [ns][rows][n] outputs of a pass"""
    lds = {_ld(t) for t in (cfs, expo) if t is not None}
    assert len(lds) <= 1, "cashflow and exposure outputs must share their leading dimension"
    return lds.pop() if lds else n


def padded_ld(n_paths: int) -> int:
    """leading dimension of a [date][state][path] tensor of n_paths paths.  A row stride that is a large power of two (2^20 paths =
    8 MiB) puts the columns of all dates and state variables a streaming kernel has in flight — 8-12 of them in the date-program
    pass — on the same HBM channel and bank for a given path: measured 0.410 ms at ld = 2^20 against 0.351 ms at 2^20 + 512 for the
    same 2^20-path pass (tools/gpu_r3w.sh).  512 doubles (4 KiB) are added whenever the stride is a multiple of 32 KiB."""
    return n_paths + 512 if n_paths >= 16384 and n_paths % 4096 == 0 else n_paths


class HipBackend:
    """One per process / GPU. Methods mirror the C ABI one-to-one; tensors are torch CUDA tensors (float64)."""

    name = "hip"

    def __init__(self, device_index: int | None = None):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("no AMD GPU visible (torch.cuda.is_available() is False); the MI355X path has no CPU fallback")
        if device_index is None:
            device_index = torch.cuda.current_device()
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        rc = self.lib.mcx_create(C.byref(h), C.c_int(device_index))
        if rc != 0:
            raise RuntimeError(f"mcx_create failed ({rc})")
        self.h = h
        self._keep = []

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.mcx_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- helpers -------------------------------------------------------------------------------------------------
    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.mcx_last_error(self.h)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def _stream(self) -> C.c_void_p:
        return _vp(torch.cuda.current_stream(self.device).cuda_stream)

    def empty(self, *shape, dtype=torch.float64) -> torch.Tensor:
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def zeros(self, *shape, dtype=torch.float64) -> torch.Tensor:
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def from_numpy(self, a: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    def device_info(self) -> dict:
        ncu, hbm = C.c_int32(), C.c_int64()
        name = C.create_string_buffer(256)
        self._check(self.lib.mcx_device_info(self.h, C.byref(ncu), C.byref(hbm), name, 256), "mcx_device_info")
        return {"n_cu": ncu.value, "hbm_bytes": hbm.value, "name": name.value.decode()}

    # ---- K1 ------------------------------------------------------------------------------------------------------
    def sim_create(self, plan):
        out = C.c_void_p()
        self._check(self.lib.mcx_sim_create(self.h, C.byref(plan.desc), C.byref(out)), "mcx_sim_create")
        return _Owned(out, self.lib.mcx_sim_destroy, plan)

    def empty_padded(self, *shape) -> torch.Tensor:
        """[..][rows][n_paths] view of a buffer whose rows have the padded leading dimension (padded_ld): the paths tensor, the
        exposure and cashflow matrices — every tensor the streaming kernels walk row by row"""
        n = int(shape[-1])
        return self.empty(*shape[:-1], padded_ld(n))[..., :n]

    empty_paths = empty_padded

    def generate_paths(self, sim, seed: int, path_offset: int, n_paths: int, inject_z=None, inject_u=None,
                       out: torch.Tensor | None = None, init_state: torch.Tensor | None = None) -> torch.Tensor:
        """init_state [n_state][n_paths]: start every path from its own state (mcx_generate_paths_from_state)"""
        plan = sim.plan
        if out is None:
            # (the library reads injected draws / start states with the leading dimension of the paths: those runs stay unpadded)
            pad = inject_z is None and inject_u is None and init_state is None
            out = self.empty_paths(plan.n_dates, plan.n_state, n_paths) if pad else self.empty(plan.n_dates, plan.n_state, n_paths)
        assert out.shape == (plan.n_dates, plan.n_state, n_paths)
        ld = _ld(out)
        assert ld == n_paths or (inject_z is None and inject_u is None and init_state is None)
        if inject_z is not None:
            assert inject_z.is_contiguous() and inject_z.shape == (plan.n_steps, plan.n_z, n_paths)
        if inject_u is not None:
            assert inject_u.is_contiguous() and inject_u.shape == (plan.n_steps, n_paths)
        zp, up = _vp(inject_z.data_ptr() if inject_z is not None else 0), _vp(inject_u.data_ptr() if inject_u is not None else 0)
        if init_state is not None:
            assert init_state.is_contiguous() and init_state.shape == (plan.n_state, n_paths) and init_state.is_cuda
            self._check(self.lib.mcx_generate_paths_from_state(
                self.h, sim.ptr, C.c_uint64(seed), C.c_uint64(path_offset), C.c_int64(n_paths), C.c_int64(ld),
                _vp(init_state.data_ptr()), _vp(out.data_ptr()), zp, up, self._stream()), "mcx_generate_paths_from_state")
            return out
        self._check(self.lib.mcx_generate_paths(
            self.h, sim.ptr, C.c_uint64(seed), C.c_uint64(path_offset), C.c_int64(n_paths), C.c_int64(ld),
            _vp(out.data_ptr()), zp, up, self._stream()), "mcx_generate_paths")
        return out

    def rng_draws(self, seed: int, path0: int, n: int, step: int, draw: int):
        """the RNG contract as the kernels consume it: (words uint32 [4][n], uniforms [2][n], Box-Muller pair [2][n])"""
        words = torch.empty((4, n), dtype=torch.int32, device=self.device)
        u = self.empty(2, n)
        z = self.empty(2, n)
        self._check(self.lib.mcx_rng_draws(self.h, C.c_uint64(seed), C.c_uint64(path0), C.c_int64(n), C.c_uint32(step),
                                           C.c_uint32(draw), _vp(words.data_ptr()), _vp(u.data_ptr()), _vp(z.data_ptr()),
                                           self._stream()), "mcx_rng_draws")
        return words, u, z

    # ---- multi-GPU exchange through the C ABI (RCCL; mcx/parallel.py uses torch.distributed for the same collectives) ---
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._check(self.lib.mcx_comm_unique_id(self.h, buf), "mcx_comm_unique_id")
        return buf.raw

    def comm_init(self, n_ranks: int, rank: int, uid: bytes):
        assert len(uid) == 128
        self._check(self.lib.mcx_comm_init(self.h, C.c_int32(n_ranks), C.c_int32(rank), C.c_char_p(uid)), "mcx_comm_init")

    def comm_destroy(self):
        self._check(self.lib.mcx_comm_destroy(self.h), "mcx_comm_destroy")

    def allreduce_(self, t: torch.Tensor) -> torch.Tensor:
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        self._check(self.lib.mcx_allreduce_f64(self.h, _vp(t.data_ptr()), C.c_int64(t.numel()), self._stream()), "mcx_allreduce_f64")
        return t

    def allgather(self, t: torch.Tensor, n_ranks: int) -> torch.Tensor:
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        out = torch.empty((n_ranks,) + tuple(t.shape), dtype=torch.float64, device=t.device)
        self._check(self.lib.mcx_allgather_f64(self.h, _vp(t.data_ptr()), _vp(out.data_ptr()), C.c_int64(t.numel()), self._stream()),
                    "mcx_allgather_f64")
        return out

    # ---- K2 ------------------------------------------------------------------------------------------------------
    def book_create(self, plan):
        out = C.c_void_p()
        self._check(self.lib.mcx_book_create(self.h, C.byref(plan.desc), C.byref(out)), "mcx_book_create")
        return _Owned(out, self.lib.mcx_book_destroy, plan)

    def book_set_exercise_replay(self, book, mode: int, bits: torch.Tensor | None):
        """mode 0 off / 1 record / 2 replay the exercise decisions in `bits` (uint8 [n_events][n_paths]) by K2 and K3"""
        if mode:
            assert bits.dtype == torch.uint8 and bits.is_cuda and bits.is_contiguous() and bits.shape[0] == len(book.plan.events)
        self._check(self.lib.mcx_book_set_exercise_replay(self.h, book.ptr, C.c_int32(mode), _vp(bits.data_ptr() if mode else 0),
                                                          C.c_int64(bits.shape[0] if mode else 0), C.c_int64(bits.shape[1] if mode else 0)),
                    "mcx_book_set_exercise_replay")
        self._keep_replay = bits if mode else None

    def new_exercise_bits(self, n_events: int, n_paths: int) -> torch.Tensor:
        return torch.zeros((n_events, n_paths), dtype=torch.uint8, device=self.device)

    def book_reset_coeffs(self, book, values: np.ndarray):
        """the coefficient array as it was uploaded at book_create (a re-run of a cached book starts from the same state)"""
        self.book_set_coeffs(book, 0, values)

    def book_set_coeffs(self, book, offset: int, values: np.ndarray):
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        book.plan.coeffs[offset:offset + v.size] = v
        self._check(self.lib.mcx_book_set_coeffs(self.h, book.ptr, C.c_int64(offset), C.c_int64(v.size), _abi.ptr(v),
                                                 self._stream()), "mcx_book_set_coeffs")

    def book_set_bridge_rng(self, book, seed: int, path_offset: int, inject: dict | None = None):
        """inject: {product index: tensor [2 * n_intervals][n_paths]} of recorded uniforms (parity runs) or None (Philox)"""
        table, ld = None, 0
        if inject:
            arr = (C.c_void_p * book.plan.desc.n_products)()
            for p_i, t in inject.items():
                assert t.is_contiguous() and t.dtype == torch.float64
                arr[p_i] = t.data_ptr()
                ld = t.shape[1]
            table = arr
        self._check(self.lib.mcx_book_set_bridge_rng(self.h, book.ptr, C.c_uint64(seed), C.c_uint64(path_offset), table,
                                                     C.c_int64(ld), self._stream()), "mcx_book_set_bridge_rng")

    def eval_book(self, book, paths: torch.Tensor):
        plan = book.plan
        n = paths.shape[2]
        cfs = self.empty_padded(plan.n_netting_sets, n) if plan.desc.want_cfs else None
        expo = self.empty_padded(plan.n_netting_sets, plan.n_expo_rows, n) if plan.desc.want_expo else None
        self._check(self.lib.mcx_eval_book(
            self.h, book.ptr, _vp(paths.data_ptr()), C.c_int64(n), C.c_int64(_ld(paths)),
            _vp(cfs.data_ptr() if cfs is not None else 0), _vp(expo.data_ptr() if expo is not None else 0),
            C.c_int64(_ld_out(cfs, expo, n)), self._stream()), "mcx_eval_book")
        return cfs, expo

    def resolve_atoms(self, book, atom_ids, paths: torch.Tensor) -> torch.Tensor:
        ids = np.ascontiguousarray(atom_ids, dtype=np.int32)
        n = paths.shape[2]
        out = self.empty(len(ids), n)
        self._check(self.lib.mcx_resolve_atoms(self.h, book.ptr, _abi.ptr(ids), C.c_int32(len(ids)),
                                               _vp(paths.data_ptr()), C.c_int64(n), C.c_int64(_ld(paths)), _vp(out.data_ptr()),
                                               C.c_int64(n), self._stream()), "mcx_resolve_atoms")
        return out

    # ---- fused pass ----------------------------------------------------------------------------------------------
    def fused_create(self, sim, book, plan):
        """returns None when the book is not fusable (the caller then runs K1/K2/K4 separately)"""
        out = C.c_void_p()
        rc = self.lib.mcx_fused_create(self.h, sim.ptr, book.ptr, C.byref(plan.desc), C.byref(out))
        if rc == _abi.E_NOT_FUSABLE:
            self.not_fusable_reason = self.lib.mcx_last_error(self.h).decode()
            return None
        self._check(rc, "mcx_fused_create")
        assert self.lib.mcx_fused_num_records(out) == plan.n_records
        f = _Owned(out, self.lib.mcx_fused_destroy, plan)
        f.sim, f.book = sim, book
        return f

    def fused_run(self, fused, seed: int, path_offset: int, n_paths: int, paths=None, cfs=None, expo=None,
                  inject_z=None, inject_u=None, device_records: bool = False, records_out=None):
        """records as a host structured array, or (device_records) as a device tensor [n_records][4] without synchronising
        (records_out: a caller-owned tensor of that shape to write them to)"""
        dp = lambda t: _vp(t.data_ptr() if t is not None else 0)
        ld_p = _ld(paths) if paths is not None else n_paths
        assert ld_p == n_paths or (inject_z is None and inject_u is None)      # (injected draws are read with the paths' leading dimension)
        if device_records:
            rec = records_out if records_out is not None else self.empty(fused.plan.n_records, 4)
            self._check(self.lib.mcx_fused_run_device(
                self.h, fused.ptr, C.c_uint64(seed), C.c_uint64(path_offset), C.c_int64(n_paths), dp(paths), C.c_int64(ld_p),
                dp(cfs), dp(expo), C.c_int64(_ld_out(cfs, expo, n_paths)), dp(inject_z), dp(inject_u), _vp(rec.data_ptr()), self._stream()),
                "mcx_fused_run_device")
            return rec
        out = np.zeros(fused.plan.n_records, dtype=_abi.ACC_DTYPE)
        self._check(self.lib.mcx_fused_run(
            self.h, fused.ptr, C.c_uint64(seed), C.c_uint64(path_offset), C.c_int64(n_paths), dp(paths), C.c_int64(ld_p),
            dp(cfs), dp(expo), C.c_int64(_ld_out(cfs, expo, n_paths)), dp(inject_z), dp(inject_u), _abi.ptr(out), self._stream()), "mcx_fused_run")
        return out

    def box_muller(self, words: torch.Tensor, table_bits: int = 7):
        """words: int32/uint32 device tensor [4][n] of Philox output blocks -> (uniforms [2][n], normals [2][n]) as the kernels map them"""
        n = words.shape[1]
        u, z = self.empty(2, n), self.empty(2, n)
        self._check(self.lib.mcx_box_muller(self.h, _vp(words.data_ptr()), C.c_int64(n), C.c_int32(table_bits), _vp(u.data_ptr()),
                                            _vp(z.data_ptr()), self._stream()), "mcx_box_muller")
        return u, z

    def fused_set_timing(self, f, every: int | bool):
        """arm (time every `every`-th launch; True = all) / disarm (0 / False) the event pairs around the main kernel of a fused pass"""
        self._check(self.lib.mcx_fused_set_timing(f.ptr, C.c_int32(int(every))), "mcx_fused_set_timing")

    def fused_kernel_times(self, f) -> np.ndarray:
        """durations (ms) of the main kernel of the passes launched since the last call (at most 64), in launch order"""
        out = np.zeros(64, dtype=np.float32)
        n = C.c_int32(0)
        self._check(self.lib.mcx_fused_kernel_times(f.ptr, _abi.ptr(out), C.c_int32(64), C.byref(n)), "mcx_fused_kernel_times")
        return out[:n.value].astype(np.float64)

    def fused_is_straight_line(self, f) -> bool:
        return bool(self.lib.mcx_fused_is_straight_line(f.ptr))

    def fused_eval_paths(self, fused, paths: torch.Tensor, cfs=None, expo=None, device_records: bool = False):
        n = paths.shape[2]
        dp = lambda t: _vp(t.data_ptr() if t is not None else 0)
        if device_records:
            rec = self.empty(fused.plan.n_records, 4)
            self._check(self.lib.mcx_fused_eval_paths_device(self.h, fused.ptr, dp(paths), C.c_int64(n), C.c_int64(_ld(paths)), dp(cfs), dp(expo),
                                                             C.c_int64(_ld_out(cfs, expo, n)), _vp(rec.data_ptr()), self._stream()), "mcx_fused_eval_paths_device")
            return rec
        out = np.zeros(fused.plan.n_records, dtype=_abi.ACC_DTYPE)
        self._check(self.lib.mcx_fused_eval_paths(self.h, fused.ptr, dp(paths), C.c_int64(n), C.c_int64(_ld(paths)), dp(cfs), dp(expo),
                                                  C.c_int64(_ld_out(cfs, expo, n)), _abi.ptr(out), self._stream()), "mcx_fused_eval_paths")
        return out

    # ---- tangents ------------------------------------------------------------------------------------------------
    def tangent_european(self, sim, opts, n_ns: int, n_params: int, seed: int, path_offset: int, n_paths: int,
                         inject_z=None, inject_u=None):
        cfs = self.empty(n_ns, n_paths)
        dcfs = self.empty(n_ns, n_params, n_paths)
        dp = lambda t: _vp(t.data_ptr() if t is not None else 0)
        self._check(self.lib.mcx_tangent_european(
            self.h, sim.ptr, opts, C.c_int32(len(opts)), C.c_int32(n_ns), C.c_uint64(seed), C.c_uint64(path_offset),
            C.c_int64(n_paths), C.c_int64(n_paths), dp(cfs), dp(dcfs), C.c_int64(n_paths), dp(inject_z), dp(inject_u),
            self._stream()), "mcx_tangent_european")
        return cfs, dcfs

    # ---- forward-mode pass through the exposure path (csrc/kt_book.hip) ----------------------------------------------
    def tangent_paths(self, sim, dslot: np.ndarray, dinit: np.ndarray, daux: np.ndarray, seed: int, path_offset: int,
                      n_paths: int, inject_z=None):
        plan = sim.plan
        NP = _abi.TANGENT_NP
        paths = self.empty(plan.n_dates, plan.n_state, n_paths)
        dpaths = self.empty(NP, plan.n_dates, plan.n_state, n_paths)
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (dslot, dinit, daux)]
        assert a[0].shape == (plan.n_slots, _abi.SLOT_NPARAM, NP) and a[1].shape == (plan.n_state, NP)
        assert a[2].shape == (plan.n_steps, plan.n_slots, _abi.AUX, NP)
        self._check(self.lib.mcx_tangent_paths(
            self.h, sim.ptr, _abi.ptr(a[0]), _abi.ptr(a[1]), _abi.ptr(a[2]), C.c_uint64(seed), C.c_uint64(path_offset),
            C.c_int64(n_paths), _vp(paths.data_ptr()), _vp(dpaths.data_ptr()), C.c_int64(n_paths),
            _vp(inject_z.data_ptr() if inject_z is not None else 0), self._stream()), "mcx_tangent_paths")
        return paths, dpaths

    def tangent_lsm(self, book, product: int, first_event: int, num_atom: int, x_atom: int, shift: float, scale: float,
                    datoms: torch.Tensor, paths: torch.Tensor, dpaths: torch.Tensor) -> np.ndarray:
        K = book.plan.n_basis
        out = np.zeros((1 + _abi.TANGENT_NP, (2 * K - 1) + K))
        n = paths.shape[2]
        self._check(self.lib.mcx_tangent_lsm(
            self.h, book.ptr, C.c_int32(product), C.c_int32(first_event), C.c_int32(num_atom), C.c_int32(x_atom),
            C.c_double(shift), C.c_double(scale), _vp(datoms.data_ptr()), _vp(paths.data_ptr()), _vp(dpaths.data_ptr()),
            C.c_int64(n), C.c_int64(n), C.c_int32(paths.shape[0]), _abi.ptr(out), self._stream()), "mcx_tangent_lsm")
        return out

    def tangent_lsm_step(self, book, product: int, roll_begin: int, roll_end: int, num_atom: int, x_atom: int, shift: float,
                         scale: float, datoms: torch.Tensor, paths: torch.Tensor, dpaths: torch.Tensor, W: torch.Tensor,
                         dW: torch.Tensor) -> np.ndarray:
        """one backward step of an exercise product's regression in dual numbers (mcx_tangent_lsm_step): W [S][n], dW [NP][S][n]
        are rolled in place; returns the moments [1+NP][(2K-1) + S K]"""
        K, S = book.plan.n_basis, W.shape[0]
        NP = _abi.TANGENT_NP
        n = paths.shape[2]
        assert W.is_contiguous() and dW.is_contiguous() and dW.shape == (NP, S, W.shape[1]) and W.shape[1] >= n
        out = np.zeros((1 + NP, (2 * K - 1) + S * K))
        self._check(self.lib.mcx_tangent_lsm_step(
            self.h, book.ptr, C.c_int32(product), C.c_int32(roll_begin), C.c_int32(roll_end), C.c_int32(num_atom), C.c_int32(x_atom),
            C.c_double(shift), C.c_double(scale), _vp(datoms.data_ptr()), _vp(paths.data_ptr()), _vp(dpaths.data_ptr()),
            C.c_int64(n), C.c_int64(n), C.c_int32(paths.shape[0]), _vp(W.data_ptr()), _vp(dW.data_ptr()), C.c_int64(W.shape[1]),
            _abi.ptr(out), self._stream()), "mcx_tangent_lsm_step")
        return out

    def tangent_eval(self, book, datoms: torch.Tensor, coeffs: torch.Tensor, dcoeffs: torch.Tensor, paths: torch.Tensor,
                     dpaths: torch.Tensor, ev_param: np.ndarray | None = None):
        plan = book.plan
        n = paths.shape[2]
        NP = _abi.TANGENT_NP
        cfs = self.empty(1 + NP, plan.n_netting_sets, n)
        expo = self.empty(1 + NP, plan.n_netting_sets, max(plan.n_expo_rows, 1), n)
        self._check(self.lib.mcx_tangent_eval(
            self.h, book.ptr, _vp(datoms.data_ptr()), _vp(coeffs.data_ptr()), _vp(dcoeffs.data_ptr()), _vp(paths.data_ptr()),
            _vp(dpaths.data_ptr()), C.c_int64(n), C.c_int64(n), C.c_int32(paths.shape[0]), _vp(cfs.data_ptr()),
            _vp(expo.data_ptr()), _abi.ptr(np.ascontiguousarray(ev_param, dtype=np.int32)) if ev_param is not None else None,
            self._stream()), "mcx_tangent_eval")
        return cfs, expo

    def tangent_pick(self, rows, threshold: float, targets, expo: torch.Tensor, ns_i: int, delayed=None, collateralized=False) -> np.ndarray:
        """-> [n_dates][1+NP]: (local index of the path realising the order statistic or -1, its tangent)"""
        r = np.ascontiguousarray(rows, dtype=np.int32)
        dl = None if delayed is None else np.ascontiguousarray(delayed, dtype=np.int32)
        tg = np.ascontiguousarray(targets, dtype=np.float64)
        out = np.zeros((len(r), 1 + _abi.TANGENT_NP))
        n = expo.shape[3]
        stride = expo.shape[1] * expo.shape[2] * expo.shape[3]
        self._check(self.lib.mcx_tangent_pick(self.h, _abi.ptr(r), _abi.ptr(dl) if dl is not None else None, C.c_int32(int(collateralized)),
                                              C.c_int32(len(r)), C.c_double(threshold), _abi.ptr(tg), _vp(expo[0, ns_i].data_ptr()),
                                              C.c_int64(stride), C.c_int64(n), C.c_int64(n), _abi.ptr(out), self._stream()), "mcx_tangent_pick")
        return out

    def tangent_cva(self, book, datoms: torch.Tensor, rows, surv, cond, threshold: float, recovery: float, expo: torch.Tensor,
                    ns_i: int, paths: torch.Tensor, dpaths: torch.Tensor, delayed=None, collateralized=False) -> torch.Tensor:
        n = paths.shape[2]
        dl = None if delayed is None else np.ascontiguousarray(delayed, dtype=np.int32)
        r, s, c = (np.ascontiguousarray(x, dtype=np.int32) for x in (rows, surv, cond))
        out = self.empty(1 + _abi.TANGENT_NP, n)
        stride = expo.shape[1] * expo.shape[2] * expo.shape[3]
        self._check(self.lib.mcx_tangent_cva(
            self.h, book.ptr, _vp(datoms.data_ptr()), _abi.ptr(r), _abi.ptr(s), _abi.ptr(c),
            _abi.ptr(dl) if dl is not None else None, C.c_int32(int(collateralized)), C.c_int32(len(r)),
            C.c_double(threshold), C.c_double(recovery), _vp(expo[0, ns_i].data_ptr()), C.c_int64(stride), _vp(paths.data_ptr()),
            _vp(dpaths.data_ptr()), C.c_int64(n), C.c_int64(n), C.c_int32(paths.shape[0]), _vp(out.data_ptr()), self._stream()),
            "mcx_tangent_cva")
        return out

    def tangent_profiles(self, rows, threshold: float, expo: torch.Tensor, ns_i: int, delayed=None, collateralized=False) -> np.ndarray:
        """-> [n_dates][2][NP] local sums of 1[u>0] du (EPE) and 1[u<0] du (ENE)"""
        r = np.ascontiguousarray(rows, dtype=np.int32)
        dl = None if delayed is None else np.ascontiguousarray(delayed, dtype=np.int32)
        out = np.zeros((len(r), 2, _abi.TANGENT_NP))
        n = expo.shape[3]
        stride = expo.shape[1] * expo.shape[2] * expo.shape[3]
        self._check(self.lib.mcx_tangent_profiles(self.h, _abi.ptr(r), _abi.ptr(dl) if dl is not None else None,
                                                  C.c_int32(int(collateralized)), C.c_int32(len(r)), C.c_double(threshold),
                                                  _vp(expo[0, ns_i].data_ptr()), C.c_int64(stride), C.c_int64(n), C.c_int64(n),
                                                  _abi.ptr(out), self._stream()), "mcx_tangent_profiles")
        return out

    # ---- K3 ------------------------------------------------------------------------------------------------------
    def rows_minmax(self, x: torch.Tensor) -> np.ndarray:
        """[rows][2] (min, max) of every row of a 2-D view of `x` (last dimension = paths)"""
        n = x.shape[-1]
        rows = x.numel() // n
        ld = _ld(x)
        out = np.zeros((rows, 2))
        self._check(self.lib.mcx_rows_minmax(self.h, _vp(x.data_ptr()), C.c_int32(rows), C.c_int64(n), C.c_int64(ld), _abi.ptr(out), self._stream()),
                    "mcx_rows_minmax")
        return out

    def book_collapse_values(self, book, lo, hi, pad: float = 0.3, rel_tol: float = 1e-14, min_terms: int = 6) -> int:
        """fit + verify the value polynomials of the book's many-term events on the column ranges lo / hi [n_dates][n_state]
        (mcx_book_collapse_values); None removes them.  Returns the number of collapsed events."""
        n = C.c_int32(0)
        if lo is None:
            self._check(self.lib.mcx_book_collapse_values(self.h, book.ptr, None, None, C.c_int32(0), C.c_int32(0), C.c_double(0.0), C.c_double(1.0),
                                                          C.c_int32(2), C.byref(n), self._stream()), "mcx_book_collapse_values")
            return 0
        lo = np.ascontiguousarray(lo, dtype=np.float64); hi = np.ascontiguousarray(hi, dtype=np.float64)
        assert lo.shape == hi.shape and lo.ndim == 2
        self._check(self.lib.mcx_book_collapse_values(self.h, book.ptr, _abi.ptr(lo), _abi.ptr(hi), C.c_int32(lo.shape[0]), C.c_int32(lo.shape[1]),
                                                      C.c_double(pad), C.c_double(rel_tol), C.c_int32(min_terms), C.byref(n), self._stream()),
                    "mcx_book_collapse_values")
        return int(n.value)

    def lsm_stats(self, book, atom_ids, paths: torch.Tensor) -> np.ndarray:
        ids = np.ascontiguousarray(atom_ids, dtype=np.int32)
        n = paths.shape[2]
        out = np.zeros((len(ids), 2))
        self._check(self.lib.mcx_lsm_stats(self.h, book.ptr, _abi.ptr(ids), C.c_int32(len(ids)), _vp(paths.data_ptr()),
                                           C.c_int64(n), C.c_int64(_ld(paths)), _abi.ptr(out), self._stream()), "mcx_lsm_stats")
        return out

    def lsm_step(self, book, product: int, roll_begin: int, roll_end: int, num_atom: int, x_atom: int, shift: float,
                 scale: float, paths: torch.Tensor, W: torch.Tensor, flags: int = 0) -> torch.Tensor:
        n = paths.shape[2]
        K = book.plan.n_basis
        S = W.shape[0]
        moments = self.empty((2 * K - 1) + S * K)
        self._check(self.lib.mcx_lsm_step(
            self.h, book.ptr, C.c_int32(product), C.c_int32(roll_begin), C.c_int32(roll_end), C.c_int32(num_atom),
            C.c_int32(x_atom), C.c_double(shift), C.c_double(scale), _vp(paths.data_ptr()), C.c_int64(n), C.c_int64(_ld(paths)),
            _vp(W.data_ptr()), C.c_int64(W.shape[1]), _vp(moments.data_ptr()), C.c_int32(int(flags)), self._stream()),
            "mcx_lsm_step")
        return moments

    def lsm_run(self, book, product: int, dates: np.ndarray, paths: torch.Tensor, W: torch.Tensor, flags: int = 0):
        """the backward induction of one product without host round trips (mcx_lsm_run): dates = LSM_DATE_DTYPE array in
        induction order -> (coefficients [n_dates][S][K], status [n_dates])"""
        dates = np.ascontiguousarray(dates, dtype=_abi.LSM_DATE_DTYPE)
        n, K, S = paths.shape[2], book.plan.n_basis, W.shape[0]
        coeffs = np.zeros((len(dates), S, K))
        status = np.zeros(len(dates), dtype=np.int32)
        self._check(self.lib.mcx_lsm_run(
            self.h, book.ptr, C.c_int32(product), _abi.ptr(dates), C.c_int32(len(dates)), _vp(paths.data_ptr()), C.c_int64(n),
            C.c_int64(_ld(paths)), _vp(W.data_ptr()), C.c_int64(W.shape[1]), _abi.ptr(coeffs), _abi.ptr(status), C.c_int32(int(flags)),
            self._stream()), "mcx_lsm_run")
        return coeffs, status

    def lsm_solve(self, book, product: int, moments: torch.Tensor, date: np.ndarray, date_index: int, table: torch.Tensor,
                  status: torch.Tensor):
        """K x K solve + coefficient scatter of one LSM date on the device (moments: device tensor, e.g. all-reduced over the ranks)"""
        d = np.ascontiguousarray(date, dtype=_abi.LSM_DATE_DTYPE).reshape(1)
        self._check(self.lib.mcx_lsm_solve(self.h, book.ptr, C.c_int32(product), _vp(moments.data_ptr()), _abi.ptr(d), C.c_int32(date_index),
                                           _vp(table.data_ptr()), _vp(status.data_ptr()), self._stream()), "mcx_lsm_solve")

    def lsm_step_batch(self, book, jobs: np.ndarray, n_states: int, paths: torch.Tensor, W: torch.Tensor, ld_w: int,
                       flags: int = 0) -> np.ndarray:
        """jobs: LSM_JOB_DTYPE array (same number of exercise states); W: the flat cashflow-cache tensor; -> moments [n_jobs][NM]"""
        jobs = np.ascontiguousarray(jobs, dtype=_abi.LSM_JOB_DTYPE)
        n = paths.shape[2]
        K = book.plan.n_basis
        out = np.zeros((len(jobs), (2 * K - 1) + n_states * K))
        self._check(self.lib.mcx_lsm_step_batch(
            self.h, book.ptr, _abi.ptr(jobs), C.c_int32(len(jobs)), C.c_int32(n_states), _vp(paths.data_ptr()), C.c_int64(n),
            C.c_int64(_ld(paths)), _vp(W.data_ptr()), C.c_int64(ld_w), C.c_int64(W.numel()), _abi.ptr(out), C.c_int32(int(flags)), self._stream()),
            "mcx_lsm_step_batch")
        return out

    def lsm_step_batch_dev(self, book, jobs: np.ndarray, n_states: int, paths: torch.Tensor, W: torch.Tensor, ld_w: int,
                           flags: int = 0) -> torch.Tensor:
        """lsm_step_batch with the moments left on the device ([n_jobs][NM] tensor, stream-ordered)"""
        jobs = np.ascontiguousarray(jobs, dtype=_abi.LSM_JOB_DTYPE)
        n = paths.shape[2]
        K = book.plan.n_basis
        out = self.empty(len(jobs), (2 * K - 1) + n_states * K)
        self._check(self.lib.mcx_lsm_step_batch_dev(
            self.h, book.ptr, _abi.ptr(jobs), C.c_int32(len(jobs)), C.c_int32(n_states), _vp(paths.data_ptr()), C.c_int64(n),
            C.c_int64(_ld(paths)), _vp(W.data_ptr()), C.c_int64(ld_w), C.c_int64(W.numel()), _vp(out.data_ptr()), C.c_int32(int(flags)),
            self._stream()), "mcx_lsm_step_batch_dev")
        return out

    def lsm_solve_batch(self, book, solve_jobs: np.ndarray, n_states: int, moments: torch.Tensor, flag: torch.Tensor):
        """K x K solves + coefficient scatter of a batched step on the device; flag: int32 device tensor [1], set on a singular system"""
        sj = np.ascontiguousarray(solve_jobs, dtype=_abi.LSM_SOLVE_JOB_DTYPE)
        self._check(self.lib.mcx_lsm_solve_batch(self.h, book.ptr, _abi.ptr(sj), C.c_int32(len(sj)), C.c_int32(n_states),
                                                 _vp(moments.data_ptr()), _vp(flag.data_ptr()), self._stream()), "mcx_lsm_solve_batch")

    def lsm_run_batch(self, book, jobs: np.ndarray, solve_jobs: np.ndarray, step_begin: np.ndarray, step_states: np.ndarray,
                      paths: torch.Tensor, W: torch.Tensor, ld_w: int, flags: int = 0) -> int:
        """every (step, solve) pair of a product-batched backward induction in one call (mcx_lsm_run_batch): the tables of ALL steps,
        step t = jobs[step_begin[t]:step_begin[t+1]] with step_states[t] exercise states -> singular-system flag"""
        jobs = np.ascontiguousarray(jobs, dtype=_abi.LSM_JOB_DTYPE)
        sj = np.ascontiguousarray(solve_jobs, dtype=_abi.LSM_SOLVE_JOB_DTYPE)
        sb = np.ascontiguousarray(step_begin, dtype=np.int32)
        ss = np.ascontiguousarray(step_states, dtype=np.int32)
        assert len(sb) == len(ss) + 1 and len(jobs) == len(sj) == int(sb[-1])
        n = paths.shape[2]
        flag = C.c_int32(0)
        self._check(self.lib.mcx_lsm_run_batch(
            self.h, book.ptr, _abi.ptr(jobs), _abi.ptr(sj), _abi.ptr(sb), _abi.ptr(ss), C.c_int32(len(ss)), _vp(paths.data_ptr()),
            C.c_int64(n), C.c_int64(_ld(paths)), _vp(W.data_ptr()), C.c_int64(ld_w), C.c_int64(W.numel()), C.byref(flag), C.c_int32(int(flags)),
            self._stream()), "mcx_lsm_run_batch")
        return int(flag.value)

    def book_get_coeffs(self, book) -> np.ndarray:
        out = np.zeros(len(book.plan.coeffs))
        self._check(self.lib.mcx_book_get_coeffs(self.h, book.ptr, C.c_int64(0), C.c_int64(len(out)), _abi.ptr(out), self._stream()),
                    "mcx_book_get_coeffs")
        return out

    def book_set_coeffs_batch(self, book, offsets: np.ndarray, values: np.ndarray):
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        v = np.ascontiguousarray(values, dtype=np.float64)
        assert v.ndim == 2 and v.shape[0] == len(off)
        self._check(self.lib.mcx_book_set_coeffs_batch(self.h, book.ptr, _abi.ptr(off), C.c_int32(len(off)), C.c_int32(v.shape[1]),
                                                       _abi.ptr(v), self._stream()), "mcx_book_set_coeffs_batch")

    # ---- K4 / K5 -------------------------------------------------------------------------------------------------
    def reduce_vector(self, x: torch.Tensor) -> np.ndarray:
        out = np.zeros(1, dtype=_abi.ACC_DTYPE)
        self._check(self.lib.mcx_reduce_vector(self.h, _vp(x.data_ptr()), C.c_int64(x.shape[0]), _abi.ptr(out),
                                               self._stream()), "mcx_reduce_vector")
        return out

    def reduce_profiles(self, unsec, expo_ns: torch.Tensor) -> np.ndarray:
        n = expo_ns.shape[1]
        unsec.desc.n_rows = expo_ns.shape[0]           # the library checks every row / delayed index against the block it is given
        out = np.zeros((unsec.n_dates, 2), dtype=_abi.ACC_DTYPE)
        self._check(self.lib.mcx_reduce_profiles(self.h, C.byref(unsec.desc), _vp(expo_ns.data_ptr()), C.c_int64(n),
                                                 C.c_int64(_ld(expo_ns)), _abi.ptr(out), self._stream()), "mcx_reduce_profiles")
        return out

    def reduce_cva(self, book, unsec, surv_atoms, cond_atoms, recovery: float, expo_ns: torch.Tensor,
                   paths: torch.Tensor) -> np.ndarray:
        n = expo_ns.shape[1]
        unsec.desc.n_rows = expo_ns.shape[0]
        sa = np.ascontiguousarray(surv_atoms, dtype=np.int32)
        ca = np.ascontiguousarray(cond_atoms, dtype=np.int32)
        out = np.zeros(1, dtype=_abi.ACC_DTYPE)
        self._check(self.lib.mcx_reduce_cva(
            self.h, book.ptr, C.byref(unsec.desc), _abi.ptr(sa), _abi.ptr(ca), C.c_double(recovery),
            _vp(expo_ns.data_ptr()), _vp(paths.data_ptr()), C.c_int64(n), C.c_int64(_ld(expo_ns)), C.c_int64(_ld(paths)),
            _abi.ptr(out), self._stream()), "mcx_reduce_cva")
        return out

    def unsecured(self, unsec, expo_ns: torch.Tensor) -> torch.Tensor:
        n = expo_ns.shape[1]
        unsec.desc.n_rows = expo_ns.shape[0]
        out = self.empty(unsec.n_dates, n)
        self._check(self.lib.mcx_unsecured(self.h, C.byref(unsec.desc), _vp(expo_ns.data_ptr()), C.c_int64(n),
                                           C.c_int64(_ld(expo_ns)), _vp(out.data_ptr()), C.c_int64(n), self._stream()),
                    "mcx_unsecured")
        return out

    def select_hist(self, unsec, expo_ns: torch.Tensor, n_sel: int, prefix: np.ndarray, shift: int, bits: int) -> torch.Tensor:
        n = expo_ns.shape[1]
        unsec.desc.n_rows = expo_ns.shape[0]
        pf = np.ascontiguousarray(prefix, dtype=np.uint64)
        hist = self.empty(unsec.n_dates, n_sel, 1 << bits, dtype=torch.int64)
        self._check(self.lib.mcx_select_hist(self.h, C.byref(unsec.desc), _vp(expo_ns.data_ptr()), C.c_int64(n),
                                             C.c_int64(_ld(expo_ns)), C.c_int32(n_sel), _abi.ptr(pf), C.c_int32(shift),
                                             C.c_int32(bits), _vp(hist.data_ptr()), self._stream()), "mcx_select_hist")
        return hist


    # device-resident select (no host round trip per digit pass)
    def select_hist_dev(self, unsec, expo_ns: torch.Tensor, n_sel: int, prefix: torch.Tensor, shift: int, bits: int, out: torch.Tensor,
                        n_paths: int | None = None):
        """n_paths < expo_ns.shape[1]: the pass over a prefix of the paths (the sample of the bracket select)"""
        ld = _ld(expo_ns)
        n = expo_ns.shape[1] if n_paths is None else min(int(n_paths), expo_ns.shape[1])
        unsec.desc.n_rows = expo_ns.shape[0]
        self._check(self.lib.mcx_select_hist_dev(self.h, C.byref(unsec.desc), _vp(expo_ns.data_ptr()), C.c_int64(n), C.c_int64(ld),
                                                 C.c_int32(n_sel), _vp(prefix.data_ptr()), C.c_int32(shift), C.c_int32(bits),
                                                 _vp(out.data_ptr()), self._stream()), "mcx_select_hist_dev")
        return out

    def select_bracket(self, unsec, expo_ns: torch.Tensor, lo: torch.Tensor, hi: torch.Tensor, counts: torch.Tensor, cand: torch.Tensor):
        """one pass over the exposures: counts[0][m] = paths below lo[m], counts[1][m] = paths inside [lo[m], hi[m]], gathered into
        cand[m] (mcx_select_bracket); counts is an int64 [2][n_dates] device tensor, cand float64 [n_dates][cap]"""
        n = expo_ns.shape[1]
        unsec.desc.n_rows = expo_ns.shape[0]
        assert counts.dtype == torch.int64 and counts.shape == (2, unsec.n_dates) and counts.is_contiguous() and cand.is_contiguous()
        self._check(self.lib.mcx_select_bracket(self.h, C.byref(unsec.desc), _vp(expo_ns.data_ptr()), C.c_int64(n), C.c_int64(_ld(expo_ns)),
                                                _vp(lo.data_ptr()), _vp(hi.data_ptr()), _vp(counts[0].data_ptr()), _vp(counts[1].data_ptr()),
                                                _vp(cand.data_ptr()), C.c_int64(cand.shape[1]), self._stream()), "mcx_select_bracket")

    def select_hist_rows(self, rows: torch.Tensor, row_n: torch.Tensor, n_sel: int, prefix: torch.Tensor, shift: int, bits: int, out: torch.Tensor):
        """digit pass over a plain [n_rows][ld] tensor whose row m holds row_n[m] values (mcx_select_hist_rows)"""
        assert rows.is_contiguous() and row_n.dtype == torch.int64
        self._check(self.lib.mcx_select_hist_rows(self.h, _vp(rows.data_ptr()), C.c_int32(rows.shape[0]), C.c_int64(rows.shape[1]),
                                                  _vp(row_n.data_ptr()), C.c_int32(n_sel), _vp(prefix.data_ptr()), C.c_int32(shift),
                                                  C.c_int32(bits), _vp(out.data_ptr()), self._stream()), "mcx_select_hist_rows")
        return out

    def select_narrow(self, hist: torch.Tensor, n_dates: int, n_sel: int, shift: int, bits: int, prefix: torch.Tensor, rem: torch.Tensor):
        self._check(self.lib.mcx_select_narrow(self.h, C.c_int32(n_dates), C.c_int32(n_sel), _vp(hist.data_ptr()), C.c_int32(shift),
                                               C.c_int32(bits), _vp(prefix.data_ptr()), _vp(rem.data_ptr()), self._stream()),
                    "mcx_select_narrow")


class _Owned:
    """native object + the plan whose host arrays it was built from"""

    def __init__(self, ptr, destroy, plan):
        self.ptr, self._destroy, self.plan = ptr, destroy, plan

    def __del__(self):
        try:
            if self.ptr:
                self._destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


_default_backend = None


def get_backend():
    """the process-wide HIP backend (created on first use; raises if the extension or the GPU is missing)"""
    global _default_backend
    if _default_backend is None:
        _default_backend = HipBackend()
    return _default_backend


def set_backend(backend):
    """tests install the CPU oracle here as the CHECKER of host logic; the product never does."""
    global _default_backend
    _default_backend = backend
