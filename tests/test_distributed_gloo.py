"""N > 1 path: world_size-2 gloo run on CPU (oracle backend as the checker).  Sharded paths + the accumulator gather,
LSM moment all-reduce and radix-select histogram all-reduce must reproduce the single-process result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases


def _worker(rank, world, port, name, out, n_main=None, n_pre=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_backend import OracleBackend
        sc, _ = cases.make_controller(name, OracleBackend(), inject=False)
        if n_main is not None:
            sc.num_paths_mainsim, sc.num_paths_presim = n_main, n_pre
        res = sc.run_simulation()
        if rank == 0:
            out.put([[[(float(v), float(e)) for v, e in m] for m in ns] for ns in res.results])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["irs_cva", "bermudan_swaption", "netting", "mixed_book_multi"])   # last: product-batched LSM
def test_two_ranks_match_single_process(name, oracle):
    single, _ = cases.make_controller(name, oracle, inject=False)
    ref = single.run_simulation().results
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + hash(name) % 50
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for ns_r, ns_g in zip(ref, got):
        for m_r, m_g in zip(ns_r, ns_g):
            a = np.array([[float(v), float(e)] for v, e in m_r])
            b = np.array(m_g)
            # same global Philox counters -> same paths; only summation order differs
            assert np.allclose(a[:, 0], b[:, 0], rtol=1e-10, atol=1e-13), (name, a[:, 0], b[:, 0])
            assert np.allclose(a[:, 1], b[:, 1], rtol=1e-7, atol=1e-12), (name, a[:, 1], b[:, 1])


def test_four_ranks_with_an_uneven_split_match_single_process(oracle):
    """4 ranks, path counts that do not divide by 4 (1001 main / 1003 pre-simulation paths: shards of 251, 250, 250, 250):
    the remainder goes to the first ranks (Shard.split), global Philox counters keep every path identical"""
    name, n_main, n_pre = "irs_cva", 1001, 1003
    single, _ = cases.make_controller(name, oracle, inject=False)
    single.num_paths_mainsim, single.num_paths_presim = n_main, n_pre
    ref = single.run_simulation().results
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 4, port, name, q, n_main, n_pre)) for r in range(4)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a = np.array([[float(v), float(e)] for v, e in ref[0][0]])
    b = np.array(got[0][0])
    assert np.allclose(a[:, 0], b[:, 0], rtol=1e-10, atol=1e-13), (a, b)
    assert np.allclose(a[:, 1], b[:, 1], rtol=1e-7, atol=1e-12), (a, b)
