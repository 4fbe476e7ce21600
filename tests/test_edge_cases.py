"""Edge cases the reference's tests touch (ragged path counts, a single path -> NaN error (metric.py:33), products past
maturity -> zero exposure (controller.py:321-322), exposure dates before the first cashflow)."""
import math

import numpy as np
import pytest

import cases
from mcx.controller.controller import SimulationController
from mcx.metrics.epe_metric import EPEMetric
from mcx.metrics.pfe_metric import PFEMetric
from mcx.metrics.pv_metric import PVMetric
from mcx.metrics.risk_metrics import RiskMetrics
from mcx.models.vasicek import VasicekModel
from mcx.products.bond import Bond
from mcx.products.netting_set import NettingSet


def _bond_case(backend, n_main, n_pre, fused=True):
    model = VasicekModel(0.0, 0.02, 0.04, 0.3, 0.015, asset_id="r")
    bond = Bond(0.0, 1.0, 1.0, 0.5, True, 0.03, "r")
    ns = [NettingSet(name="b", products=[bond])]
    rm = RiskMetrics([EPEMetric(), PFEMetric(0.95), PVMetric()], exposure_timeline=np.array([0.0, 0.5, 1.0, 1.5]))
    sc = SimulationController(ns, model, rm, n_main, n_pre, 3, cases.E, backend=backend)
    sc.allow_fused = fused
    return sc, sc.run_simulation()


def test_single_path_gives_nan_error(oracle):
    sc, res = _bond_case(oracle, 1, 64)
    pv, err = res.results[0][2][0]
    assert math.isfinite(pv) and math.isnan(err)


def test_exposure_after_maturity_is_zero(oracle):
    sc, res = _bond_case(oracle, 300, 300)
    epe = [v for v, _ in res.results[0][0]]
    assert epe[2] == 0.0 and epe[3] == 0.0           # t = 1.0 (cashflow of that date excluded) and t = 1.5 (past maturity)
    assert epe[0] > 0.9 and epe[1] > 0.9


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 64, 257, 1000, 4097])
@pytest.mark.parametrize("fused", [True, False])
def test_ragged_path_counts_gpu_vs_oracle(n, fused, hip, oracle):
    sg, rg = _bond_case(hip, n, 300, fused)
    sc, rc = _bond_case(oracle, n, 300, False)
    for mg, mc in zip(rg.results[0], rc.results[0]):
        a, b = np.array(mg, dtype=float), np.array(mc, dtype=float)
        assert np.allclose(a[:, 0], b[:, 0], rtol=1e-9, atol=1e-12, equal_nan=True), (n, a, b)
        if n > 1:
            assert np.allclose(a[:, 1], b[:, 1], rtol=1e-6, atol=1e-12, equal_nan=True), (n, a, b)


def test_rerun_reuses_the_compiled_book_and_recompiles_after_a_parameter_change(oracle):
    """with reuse_compiled (opt-in) a second run_simulation() of the same controller skips the host compilation (descriptor cache
    keyed on the model parameter values) and reproduces the first run bit for bit; changing a parameter, or invalidate() after a
    product mutation, recompiles.  The default recompiles on every run like the reference (controller.py:663-709)."""
    import torch
    sc, _ = cases.make_controller("bermudan_swaption", oracle)
    r1 = sc.run_simulation()
    book0 = sc.book
    sc.run_simulation()
    assert sc.book is not book0                       # default: nothing is carried over
    sc.reuse_compiled = True
    r1 = sc.run_simulation()
    book1 = sc.book
    r2 = sc.run_simulation()
    assert sc.book is book1
    sc.invalidate()
    sc.run_simulation()
    assert sc.book is not book1
    book1 = sc.book
    for a, b in zip(r1.results[0], r2.results[0]):
        assert np.array_equal(np.array(a), np.array(b))
    with torch.no_grad():
        sc.model.get_model_params()[1].mul_(1.1)          # volatility
    r3 = sc.run_simulation()
    assert sc.book is not book1
    assert not np.allclose(np.array(r3.results[0][0])[:, 0], np.array(r1.results[0][0])[:, 0], rtol=1e-9)


def _second_order_check(backend, n_paths):
    """compute_higher_derivatives() on a Monte-Carlo PV (controller.py:253-255, 631-648): differences of the pathwise first-order
    sensitivities with common random numbers against the closed-form gamma / vomma / vanna (Monte-Carlo noise ~2 % at 200 k paths)"""
    from scipy.stats import norm
    ns, model, rm = cases.bs_european()
    sc = SimulationController(ns, model, rm, n_paths, 0, 2, cases.A, differentiate=True, backend=backend)
    sc.compute_higher_derivatives()
    res = sc.run_simulation()
    H = res.get_second_derivatives(0, "pv", evaluation_idx=0)
    S, K, r, sig, T = 120.0, 100.0, 0.05, 0.2, 2.0
    d1 = (math.log(S / K) + (r + 0.5 * sig * sig) * T) / (sig * math.sqrt(T))
    d2 = d1 - sig * math.sqrt(T)
    gamma = norm.pdf(d1) / (S * sig * math.sqrt(T))
    vega = S * norm.pdf(d1) * math.sqrt(T)
    assert abs(H["spot"]["spot"] - gamma) < 0.08 * gamma
    assert abs(H["volatility"]["volatility"] - vega * d1 * d2 / sig) < 0.05 * abs(vega * d1 * d2 / sig)
    assert abs(H["spot"]["volatility"] - (-norm.pdf(d1) * d2 / sig)) < 0.08 * abs(norm.pdf(d1) * d2 / sig)
    assert abs(H["spot"]["volatility"] - H["volatility"]["spot"]) < 0.08 * abs(H["spot"]["volatility"])          # symmetric up to noise
    assert res.get_derivatives(0, "pv", evaluation_idx=0)["spot"] == pytest.approx(0.875, abs=0.01)
    return H


def test_monte_carlo_second_order_derivatives_converge_to_the_black_scholes_hessian(oracle):
    _second_order_check(oracle, 200000)


@pytest.mark.gpu
def test_monte_carlo_second_order_derivatives_on_the_gpu(hip, oracle):
    """the same through the tangent kernel on the GPU (kt_bs: 2P + 1 first-order passes), and equal to the oracle's Hessian on the
    same Philox counters"""
    Hg = _second_order_check(hip, 200000)
    Ho = _second_order_check(oracle, 200000)
    for a in Hg:
        for b in Hg[a]:
            assert Hg[a][b] == pytest.approx(Ho[a][b], rel=1e-6, abs=1e-9), (a, b)


@pytest.mark.gpu
def test_library_rejects_indices_a_kernel_would_read_out_of_bounds(hip):
    """every row / date / offset a kernel dereferences is checked on the host (a wild index on the GPU can take the node down):
    exposure rows of the metric descriptors, atom dates of a book, cashflow-cache offsets of the batched LSM step"""
    import torch
    from mcx.plan import UnsecuredSpec
    E, n = 5, 4096
    expo = torch.zeros(E, n, dtype=torch.float64, device=hip.device)
    ok = hip.reduce_profiles(UnsecuredSpec(np.arange(E), None, 0.0, False), expo)
    assert ok.shape[0] == E
    for rows, delayed in ((np.array([0, 1, E]), None), (np.array([0, -1, 2]), None), (np.arange(3), np.array([-1, 0, E])),
                          (np.arange(3), np.array([-2, 0, 1]))):
        spec = UnsecuredSpec(rows, delayed, 0.0, delayed is not None)
        with pytest.raises(RuntimeError, match="outside the"):
            hip.reduce_profiles(spec, expo)
        with pytest.raises(RuntimeError, match="outside the"):
            hip.unsecured(spec, expo)
        with pytest.raises(RuntimeError, match="outside the"):
            hip.select_hist(spec, expo, 1, np.zeros((len(rows), 1), dtype=np.uint64), 53, 11)
    # a book whose atom reads a date the paths tensor does not have
    sc, _ = cases.make_controller("irs_cva", hip, inject=False)
    sc.prepare()
    plan = sc.book_plan
    saved = plan.atoms["t_idx"].copy()
    plan.atoms["t_idx"][0] = plan.desc.n_dates
    try:
        with pytest.raises(RuntimeError):
            hip.book_create(plan)
    finally:
        plan.atoms["t_idx"][:] = saved
    # batched LSM step: a cashflow-cache block that ends beyond the cache tensor
    from mcx import _abi
    jobs = np.zeros(1, dtype=_abi.LSM_JOB_DTYPE)
    jobs["product"], jobs["roll_begin"], jobs["roll_end"] = 0, 0, 0
    jobs["num_atom"], jobs["x_atom"] = 0, 0
    paths = sc.last_state.get("paths_pre")
    if paths is None:
        paths = torch.zeros(sc.sim_plan.n_dates, sc.sim_plan.n_state, 1024, dtype=torch.float64, device=hip.device)
    npre = paths.shape[2]
    W = torch.zeros(npre, dtype=torch.float64, device=hip.device)
    jobs["w_offset"] = 1
    with pytest.raises(RuntimeError, match="outside d_W"):
        hip.lsm_step_batch(sc.book, jobs, 1, paths, W, npre)
    # the one-call induction (mcx_lsm_run_batch) checks the same fields of every job of every step, and its step table
    sj = np.zeros(1, dtype=_abi.LSM_SOLVE_JOB_DTYPE)
    sj["coeff_off"][:] = -1
    sj["scale"] = 1.0
    with pytest.raises(RuntimeError, match="outside d_W"):
        hip.lsm_run_batch(sc.book, jobs, sj, np.array([0, 1]), np.array([1]), paths, W, npre)
    jobs["w_offset"] = 0
    sj["coeff_off"][0, 0] = len(plan.coeffs) - 1                       # a [1][K] block that ends beyond the coefficient array
    with pytest.raises(RuntimeError, match="coefficient offset"):
        hip.lsm_run_batch(sc.book, jobs, sj, np.array([0, 1]), np.array([1]), paths, W, npre)
    sj["coeff_off"][0, 0] = -1
    with pytest.raises(RuntimeError, match="states"):                  # product 0 has one exercise state
        hip.lsm_run_batch(sc.book, jobs, sj, np.array([0, 1]), np.array([2]), paths, torch.zeros(2 * npre, dtype=torch.float64, device=hip.device), npre)
    jobs["roll_end"] = 10 ** 6
    with pytest.raises(RuntimeError, match="roll window"):
        hip.lsm_run_batch(sc.book, jobs, sj, np.array([0, 1]), np.array([1]), paths, W, npre)
    jobs["roll_end"] = 0
    # the valid table runs: atom 0 is the same number on every path, i.e. a singular system unless the job says so (rank-1 solution)
    assert hip.lsm_run_batch(sc.book, jobs, sj, np.array([0, 1]), np.array([1]), paths, W, npre) == 1
    sj["degenerate"] = 1
    assert hip.lsm_run_batch(sc.book, jobs, sj, np.array([0, 1]), np.array([1]), paths, W, npre) == 0


@pytest.mark.gpu
def test_padded_leading_dimension_changes_no_number(hip):
    """the path / exposure / cashflow matrices carry a padded row stride where the path count would make it a large power of two
    (mcx._native.padded_ld): the same run on contiguous and on padded tensors, every plan, must agree bit for bit"""
    import torch
    from mcx import _native
    n = 32768
    assert _native.padded_ld(n) == n + 512 and _native.padded_ld(n + 1) == n + 1 and _native.padded_ld(4096) == 4096
    outs = {}
    for padded in (True, False):
        for plan in ("fused", "semi", "unfused"):
            build, n_pre, _n_main, steps, scheme, _diff = cases.CASES["irs_cva"]
            ns, model, rm = build()
            sc = cases.SimulationController(ns, model, rm, n, n_pre, steps, scheme, backend=hip if padded else _Unpadded(hip))
            sc.materialize, sc.allow_fused, sc.main_plan = True, plan != "unfused", ("auto" if plan == "unfused" else plan)
            res = sc.run_simulation()
            st = sc.last_state
            assert (st["paths"].stride(1) == n + 512) == padded and (st["expo"].stride(1) == n + 512) == padded
            outs[(padded, plan)] = (np.array(res.results[0][0]), st["paths"].cpu().numpy(), st["expo"].cpu().numpy())
    for plan in ("fused", "semi", "unfused"):
        for a, b in zip(outs[(True, plan)], outs[(False, plan)]):
            assert np.array_equal(a, b), plan


class _Unpadded:
    """the HIP backend with contiguous matrices (what every run used before the padded leading dimension)"""

    def __init__(self, be):
        self._be = be

    def __getattr__(self, name):
        return getattr(self._be, name)

    def empty_padded(self, *shape):
        return self._be.empty(*shape)

    empty_paths = empty_padded

    def generate_paths(self, sim, seed, path_offset, n_paths, inject_z=None, inject_u=None, out=None, init_state=None):
        if out is None:
            out = self._be.empty(sim.plan.n_dates, sim.plan.n_state, n_paths)
        return self._be.generate_paths(sim, seed, path_offset, n_paths, inject_z, inject_u, out=out, init_state=init_state)

    def eval_book(self, book, paths):
        import ctypes as C
        from mcx import _native
        be, plan, n = self._be, book.plan, paths.shape[2]
        cfs = be.empty(plan.n_netting_sets, n) if plan.desc.want_cfs else None
        expo = be.empty(plan.n_netting_sets, plan.n_expo_rows, n) if plan.desc.want_expo else None
        be._check(be.lib.mcx_eval_book(be.h, book.ptr, C.c_void_p(paths.data_ptr()), C.c_int64(n), C.c_int64(paths.stride(1)),
                                       C.c_void_p(cfs.data_ptr() if cfs is not None else 0), C.c_void_p(expo.data_ptr() if expo is not None else 0),
                                       C.c_int64(n), be._stream()), "mcx_eval_book")
        return cfs, expo


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
def test_cir_intensity_starting_at_zero(fused, hip, oracle, monkeypatch):
    """the reference asserts y0 > 0 (cirpp.py:40), and the one-launch kernel relies on it (no zero test under the diffusion root);
    a C-ABI caller may still hand over a zero start: mcx_fused_create then routes the book to the interpreter kernel, and every
    plan reproduces sqrt(clamp(0, 0)) = 0 of cirpp.py:194 (same paths and CVA as the CPU oracle)"""
    from mcx.models.cirpp import CIRPPModel
    monkeypatch.setattr(CIRPPModel, "_initial_state", lambda self: [0.0, 0.0])
    out = {}
    for be in (hip, oracle):
        sc, _ = cases.make_controller("irs_cva", be, inject=False, fused=fused and be is hip)
        res = sc.run_simulation()
        out[be.name] = (np.array(res.results[0][0][0]), sc.last_state["paths"].cpu().numpy() if be is hip else sc.last_state["paths"].numpy())
    assert np.all(np.isfinite(out["hip"][1]))
    assert np.allclose(out["hip"][1], out["oracle"][1], rtol=1e-10, atol=1e-13)
    assert np.allclose(out["hip"][0], out["oracle"][0], rtol=1e-8, atol=1e-14), (out["hip"][0], out["oracle"][0])


@pytest.mark.gpu
def test_library_rejects_scheme_without_step_map_and_short_replay_buffers(hip):
    """C-ABI callers bypass the Python guards (Model._supports_scheme): mcx_sim_create refuses a (model, scheme) pair that has no
    step map (the kernels would run on all-zero derived step constants), mcx_book_set_exercise_replay refuses a decision buffer
    with fewer rows than the book has events, and the LSM / book kernels refuse one narrower than the path count"""
    import torch
    sc, _ = cases.make_controller("irs_cva", hip, inject=False)
    sc.prepare()
    desc = sc.sim_plan.desc
    saved = desc.scheme
    try:
        for bad in (1, 3, 2):                          # MILSTEIN, QE, ANALYTICAL for Vasicek + CIR++
            desc.scheme = bad
            with pytest.raises(RuntimeError, match="no step map"):
                hip.sim_create(sc.sim_plan)
    finally:
        desc.scheme = saved
    hip.sim_create(sc.sim_plan)
    sb, _ = cases.make_controller("bermudan_swaption", hip, inject=False)
    sb.prepare()
    n_ev = len(sb.book_plan.events)
    bits = torch.zeros((n_ev - 1, 256), dtype=torch.uint8, device=hip.device)
    with pytest.raises(RuntimeError, match="rows for a book"):
        import ctypes as C
        hip._check(hip.lib.mcx_book_set_exercise_replay(hip.h, sb.book.ptr, C.c_int32(1), C.c_void_p(bits.data_ptr()),
                                                        C.c_int64(n_ev - 1), C.c_int64(256)), "mcx_book_set_exercise_replay")
    narrow = torch.zeros((n_ev, 8), dtype=torch.uint8, device=hip.device)
    hip.book_set_exercise_replay(sb.book, 1, narrow)
    try:
        paths = torch.zeros(sb.sim_plan.n_dates, sb.sim_plan.n_state, 1024, dtype=torch.float64, device=hip.device)
        with pytest.raises(RuntimeError, match="narrower than the path count"):
            hip.eval_book(sb.book, paths)
    finally:
        hip.book_set_exercise_replay(sb.book, 0, None)
