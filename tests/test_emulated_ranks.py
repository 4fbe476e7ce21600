"""The device-collective branches of the controller with MORE THAN ONE rank, on one GPU (tests/emulated_ranks.py): 4 emulated ranks
over uneven path ranges must reproduce the single-shard run — LSM coefficients, CVA, EPE / ENE, PFE order statistics.  What a
one-rank RCCL group cannot show (every collective is the identity there): a missed all-reduce of the LSM moments or of a select
histogram, a min / max range that was not gathered, records gathered in the wrong layout.  Reference: the single-process run
(controller/controller.py:677-694); SURVEY §8e."""
import threading

import numpy as np
import pytest
import torch

import cases
from emulated_ranks import EmulatedShard, EmulatedWorld, run_ranks


def test_harness_collectives_on_host_tensors():
    """the stand-in itself: sum / stack over 3 threads, repeated (slots are reused), float64 and int64"""
    world = 3
    shared = EmulatedWorld(world, timeout=30.0)
    out = [None] * world

    def work(rank):
        sh = EmulatedShard(rank, shared)
        acc = []
        for k in range(5):
            t = torch.full((4,), float(rank + 1) * (k + 1), dtype=torch.float64)
            acc.append(sh.all_reduce_(t).clone())
            acc.append(torch.from_numpy(sh.all_gather_np(np.array([rank * 10.0 + k]))))
            h = torch.full((2, 2), rank + k, dtype=torch.int64)
            acc.append(sh.all_reduce_(h).clone())
        out[rank] = acc

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(60) for t in th]
    for rank in range(world):
        assert out[rank] is not None
        for k in range(5):
            assert torch.equal(out[rank][3 * k], torch.full((4,), 6.0 * (k + 1), dtype=torch.float64))
            assert torch.equal(out[rank][3 * k + 1].flatten(), torch.tensor([0.0 + k, 10.0 + k, 20.0 + k], dtype=torch.float64))
            assert torch.equal(out[rank][3 * k + 2], torch.full((2, 2), 3 + 3 * k, dtype=torch.int64))
    assert EmulatedShard(1, EmulatedWorld(4)).split(1001) == (251, 250) and EmulatedShard(0, EmulatedWorld(4)).split(1001) == (0, 251)


def _results(res):
    return [[np.array(m, dtype=float) for m in ns] for ns in res.results]


def _coeffs(sc):
    out = []
    for p in sc.products:
        rc = getattr(p, "regression_coeffs", None)
        if rc is not None and hasattr(rc, "numpy") and rc.numel():
            out.append(rc.numpy().copy())
    for rc in getattr(sc, "regression_coeffs", []) or []:
        if rc is not None and hasattr(rc, "numpy") and rc.numel():
            out.append(rc.numpy().copy())
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name,n_main,n_pre", [("irs_cva", 40003, 20001), ("bermudan_swaption", 30001, 20003), ("netting", 20002, 8001),
                                                ("mixed_book_multi", 12001, 6001)])
def test_four_emulated_ranks_match_the_single_shard_run(name, n_main, n_pre, hip):
    from mcx import _native

    def build(be):
        sc, _ = cases.make_controller(name, be, inject=False)
        sc.materialize = False
        sc.num_paths_mainsim, sc.num_paths_presim = n_main, n_pre
        return sc

    single = build(hip)
    ref = _results(single.run_simulation())
    ref_coeffs = _coeffs(single)

    def body(sc, rank):
        res = sc.run_simulation()
        return _results(res), _coeffs(sc), sc._shard.split(n_main)

    out, calls = run_ranks(4, lambda rank: build(_native.HipBackend(0)), body)
    assert calls["all_reduce"] + calls["all_gather"] > 0
    assert sorted(o[2] for o in out) == sorted(EmulatedShard(r, EmulatedWorld(4)).split(n_main) for r in range(4))
    for rank, (got, coeffs, _) in enumerate(out):
        assert len(coeffs) == len(ref_coeffs)
        for a, b in zip(ref_coeffs, coeffs):
            assert np.allclose(a, b, rtol=1e-8, atol=1e-11), (name, rank, np.abs(a - b).max())
        for ns_r, ns_g in zip(ref, got):
            for m_r, m_g in zip(ns_r, ns_g):
                # same global Philox counters -> same paths; summation order (and the regression coefficients' last bits) differ
                assert np.allclose(m_r[:, 0], m_g[:, 0], rtol=1e-9, atol=1e-12), (name, rank, m_r[:, 0], m_g[:, 0])
                assert np.allclose(m_r[:, 1], m_g[:, 1], rtol=1e-6, atol=1e-11, equal_nan=True), (name, rank)
    for got, _, _ in out[1:]:                                             # every rank ends with the same merged result
        for ns_a, ns_b in zip(out[0][0], got):
            for a, b in zip(ns_a, ns_b):
                assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [1, 4, 7])
def test_random_rate_books_on_three_emulated_ranks(case, hip):
    """random linear books (test_hip_parity._random_rate_book: thresholds, margin periods, CVA + PV + EPE + ENE + PFE) with the paths
    on three ranks of uneven size: every metric of every netting set as on one shard"""
    from mcx import _native
    from test_hip_parity import _random_rate_book
    n_main, n_pre = 30011, 9001

    def build(be):
        ns, model, rm = _random_rate_book(case)
        sc = cases.SimulationController(ns, model, rm, n_main, n_pre, 2, cases.E, backend=be)
        sc.materialize = False
        return sc

    ref = _results(build(hip).run_simulation())
    out, calls = run_ranks(3, lambda rank: build(_native.HipBackend(0)), lambda sc, rank: _results(sc.run_simulation()))
    assert calls["all_reduce"] + calls["all_gather"] > 0
    for rank, got in enumerate(out):
        for ns_r, ns_g in zip(ref, got):
            for m_i, (m_r, m_g) in enumerate(zip(ns_r, ns_g)):
                assert np.allclose(m_r[:, 0], m_g[:, 0], rtol=1e-9, atol=1e-12), (case, rank, m_i, m_r[:, 0], m_g[:, 0])
                assert np.allclose(m_r[:, 1], m_g[:, 1], rtol=1e-6, atol=1e-11, equal_nan=True), (case, rank, m_i)


@pytest.mark.gpu
def test_pipelined_passes_gather_the_records_of_every_emulated_rank(hip):
    """bench.py's loop (fused_pass_begin / fused_pass_end: record gather on a side stream while the next kernel runs) on 3
    emulated ranks: every pass must deliver the CVA of ALL paths"""
    from mcx import _native
    n_main, n_pre = 90001, 16384

    def build(be):
        sc, _ = cases.make_controller("irs_cva", be, inject=False)
        sc.materialize = False
        sc.num_paths_mainsim, sc.num_paths_presim = n_main, n_pre
        return sc

    single = build(hip)
    single.prepare()
    ref = single.main_pass()[0][0][0]

    def body(sc, rank):
        sc.prepare()
        assert sc.pipelined_passes_available()
        vals, pending = [], None
        for _ in range(4):
            ticket = sc.fused_pass_begin()
            if pending is not None:
                vals.append(sc.fused_pass_end(pending)[0][0][0])
            pending = ticket
        vals.append(sc.fused_pass_end(pending)[0][0][0])
        return vals

    out, _ = run_ranks(3, lambda rank: build(_native.HipBackend(0)), body)
    for vals in out:
        assert len(vals) == 4
        for v, e in vals:
            assert np.isclose(v, ref[0], rtol=1e-10) and np.isclose(e, ref[1], rtol=1e-6)


@pytest.mark.gpu
def test_bracket_select_over_three_emulated_ranks(hip):
    """PFE through the one-pass bracket select (k5_bracket) with the paths on three ranks: every rank samples its own paths, the
    widest bracket is taken, `below` / `inside` counts and the candidates' digit histograms are all-reduced — the order statistics
    must equal the single-shard ones exactly given equal exposures (here: to the last bits of the regression coefficients)"""
    from mcx import _native
    n_main, n_pre = 600001, 20003

    def build(be):
        sc, _ = cases.make_controller("bermudan_swaption", be, inject=False)
        sc.materialize = False
        sc.num_paths_mainsim, sc.num_paths_presim = n_main, n_pre
        return sc

    single = build(hip)
    ref = _results(single.run_simulation())
    assert single.last_select["bracket_dates"] > 0

    def body(sc, rank):
        res = sc.run_simulation()
        return _results(res), dict(sc.last_select)

    out, calls = run_ranks(3, lambda rank: build(_native.HipBackend(0)), body)
    for got, info in out:
        assert info["bracket_dates"] > 0
        for ns_r, ns_g in zip(ref, got):
            for m_r, m_g in zip(ns_r, ns_g):
                assert np.allclose(m_r[:, 0], m_g[:, 0], rtol=1e-9, atol=1e-12)
