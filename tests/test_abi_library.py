"""The C-ABI shared library loads and exports every symbol include/mcx.h declares (no compute: runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mcx.h")).read()
    return sorted(set(re.findall(r"^(?:int|void|const char\*)\s+(mcx_[a-z0-9_]+)\s*\(", text, flags=re.M)))


def test_header_symbols_are_exported():
    from mcx import _native
    assert os.path.exists(_native.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_native.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/mcx.h but not exported"
    assert lib.mcx_abi_version() == 6
    assert set(_native._EXPORTS) == set(syms)


def test_struct_layouts_match_header():
    """sizes the kernels assume (mirrored in mcx/_abi.py)"""
    from mcx import _abi
    assert ctypes.sizeof(_abi.Slot) == 16 + 8 * _abi.SLOT_NPARAM
    assert ctypes.sizeof(_abi.SimDesc) == 40 + 8 * ctypes.sizeof(_abi.Slot) + 4 * 8
    assert ctypes.sizeof(_abi.BookDesc) == 12 * 4 + 5 * 8
    assert ctypes.sizeof(_abi.UnsecuredDesc) == 8 + 8 + 16 + 8


def test_product_fails_loudly_without_gpu():
    import pytest
    import torch
    from mcx import _native
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no AMD GPU|not built"):
        _native.HipBackend()
