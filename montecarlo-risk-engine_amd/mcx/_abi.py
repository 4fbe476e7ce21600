"""ctypes mirror of include/mcx.h (the C-ABI boundary).  Pure data-format definitions: no compute here."""
from __future__ import annotations

import ctypes as C

import numpy as np

ABI_VERSION = 6
SELECT_LOST = 1 << 44        # MCX_SELECT_LOST (mcx_select_bracket)
MAX_SLOTS = 8
MAX_Z = 8
MAX_STATE = 16
SLOT_NPARAM = 8
AUX = 8
MAX_BASIS = 6
MAX_STATES = 8

SCHEME_EULER, SCHEME_MILSTEIN, SCHEME_ANALYTICAL, SCHEME_QE = 0, 1, 2, 3
MODEL_BS, MODEL_HESTON, MODEL_VASICEK, MODEL_CIRPP, MODEL_CIRPP_DET, MODEL_HW, MODEL_S2F = 1, 2, 3, 4, 5, 6, 7
FLAG_SMOOTHING = 1
LSM_MFMA, LSM_F32_CACHE = 1, 2
EV_CASHFLOW, EV_OPTION, EV_EXERCISE, EV_EXPO_POLY, EV_EXPO_BS = 1, 2, 3, 4, 5


class Slot(C.Structure):
    _fields_ = [("kind", C.c_int32), ("state_off", C.c_int32), ("z_off", C.c_int32), ("flags", C.c_int32),
                ("p", C.c_double * SLOT_NPARAM)]


STEP_DTYPE = np.dtype([("dt", "<f8"), ("sqrt_dt", "<f8"), ("t1", "<f8"), ("store_idx", "<i4"), ("chol_idx", "<i4")])
ATOM_DTYPE = np.dtype([("t_idx", "<i4"), ("col", "<i4"), ("a", "<f8"), ("d", "<f8"), ("b", "<f8"), ("c0", "<f8"), ("c1", "<f8")])
TERM_DTYPE = np.dtype([("w", "<f8"), ("atom", "<i4"), ("den", "<i4")])
EVENT_DTYPE = np.dtype([("kind", "<i4"), ("t_idx", "<i4"), ("num_atom", "<i4"), ("x_atom", "<i4"),
                        ("term_begin", "<i4"), ("term_end", "<i4"), ("coeff_off", "<i4"), ("expo_row", "<i4"),
                        ("strike", "<f8"), ("sign", "<f8"), ("aux", "<f8", (4,))])
PRODUCT_DTYPE = np.dtype([("ev_begin", "<i4"), ("ev_end", "<i4"), ("cf_begin", "<i4"), ("cf_end", "<i4"),
                          ("netting_set", "<i4"), ("init_state", "<i4"), ("n_states", "<i4"), ("flags", "<i4")])
ACC_DTYPE = np.dtype([("n", "<f8"), ("shift", "<f8"), ("s1", "<f8"), ("s2", "<f8")])

assert STEP_DTYPE.itemsize == 32 and ATOM_DTYPE.itemsize == 48 and TERM_DTYPE.itemsize == 16
assert EVENT_DTYPE.itemsize == 80 and PRODUCT_DTYPE.itemsize == 32 and ACC_DTYPE.itemsize == 32


class SimDesc(C.Structure):
    _fields_ = [("scheme", C.c_int32), ("n_slots", C.c_int32), ("n_state", C.c_int32), ("n_z", C.c_int32),
                ("n_uniform", C.c_int32), ("n_steps", C.c_int32), ("n_dates", C.c_int32), ("n_chol", C.c_int32),
                ("n_initial_store", C.c_int32), ("flags", C.c_int32),
                ("slots", Slot * MAX_SLOTS),
                ("steps", C.c_void_p), ("chol", C.c_void_p), ("aux", C.c_void_p), ("init_state", C.c_void_p)]


class BookDesc(C.Structure):
    _fields_ = [("n_atoms", C.c_int32), ("n_terms", C.c_int32), ("n_events", C.c_int32), ("n_products", C.c_int32),
                ("n_netting_sets", C.c_int32), ("n_expo_rows", C.c_int32), ("n_basis", C.c_int32), ("n_coeffs", C.c_int32),
                ("want_cfs", C.c_int32), ("want_expo", C.c_int32), ("n_state", C.c_int32), ("n_dates", C.c_int32),
                ("atoms", C.c_void_p), ("terms", C.c_void_p), ("events", C.c_void_p), ("products", C.c_void_p),
                ("coeffs", C.c_void_p)]


class UnsecuredDesc(C.Structure):
    _fields_ = [("n_dates", C.c_int32), ("collateralized", C.c_int32), ("threshold", C.c_double),
                ("row", C.c_void_p), ("delayed", C.c_void_p), ("n_rows", C.c_int32), ("reserved", C.c_int32)]


class FusedNsDesc(C.Structure):
    _fields_ = [("netting_set", C.c_int32), ("n_dates", C.c_int32), ("want_profiles", C.c_int32), ("want_cva", C.c_int32),
                ("threshold", C.c_double), ("recovery", C.c_double),
                ("row", C.c_void_p), ("surv_atoms", C.c_void_p), ("cond_atoms", C.c_void_p)]


class FusedDesc(C.Structure):
    _fields_ = [("n_netting_sets", C.c_int32), ("want_pv", C.c_int32), ("n_expo_rows", C.c_int32), ("reserved", C.c_int32),
                ("row_t_idx", C.c_void_p), ("ns", C.c_void_p)]


E_NOT_FUSABLE = -10
FUSED_MAX_NS = 4


def ptr(a: np.ndarray | None) -> C.c_void_p:
    """host pointer of a C-contiguous numpy array (None -> NULL)"""
    if a is None:
        return C.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


# mcx_lsm_job (include/mcx.h)
LSM_DATE_DTYPE = np.dtype([("roll_begin", np.int32), ("roll_end", np.int32), ("num_atom", np.int32), ("x_atom", np.int32),
                           ("degenerate", np.int32), ("reserved", np.int32), ("coeff_off", np.int64, (2,)),
                           ("shift", np.float64), ("scale", np.float64), ("x0", np.float64)])
assert LSM_DATE_DTYPE.itemsize == 64
LSM_JOB_DTYPE = np.dtype([("product", np.int32), ("roll_begin", np.int32), ("roll_end", np.int32), ("num_atom", np.int32),
                          ("x_atom", np.int32), ("reserved", np.int32), ("w_offset", np.int64), ("shift", np.float64),
                          ("scale", np.float64)], align=True)

LSM_SOLVE_JOB_DTYPE = np.dtype([("shift", np.float64), ("scale", np.float64), ("x0", np.float64), ("coeff_off", np.int64, (2,)),
                                ("degenerate", np.int32), ("reserved", np.int32)], align=True)
assert LSM_SOLVE_JOB_DTYPE.itemsize == 48

TANGENT_NP = 4          # MCX_TANGENT_NP: model parameters per forward-mode pass (csrc/kt_book.hip)
