"""Several path-shard ranks emulated on ONE GPU, for the device-collective branches of the controller.

The build box has one GPU, and a one-rank RCCL group makes every collective the identity: a missed all-reduce of the LSM moments,
of a select histogram or of the explanatory ranges cannot show there.  Here W Python threads of one process each run a controller
on its own HIP handle (include/mcx.h: distinct handles may be used from distinct threads) over its own path range, and a `Shard`
stand-in combines the ranks' DEVICE tensors at a thread barrier — sum for all_reduce, stack for all_gather — exactly where the
production Shard (mcx/parallel.py) calls RCCL.  It reports backend "nccl" / device_collectives, so the controller takes the same
branches as on eight GPUs: mcx_lsm_step -> all-reduce -> mcx_lsm_solve, mcx_select_hist_dev -> all-reduce -> mcx_select_narrow,
device accumulator records -> all-gather, the pipelined passes' side-stream gather.
"""
import threading

import numpy as np
import torch


class EmulatedWorld:
    def __init__(self, world: int, timeout: float = 120.0):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=timeout)      # a rank that dies breaks the barrier: nobody waits forever
        self.slots = [None] * world
        self.calls = dict(all_reduce=0, all_gather=0)

    def exchange(self, rank: int, t: torch.Tensor):
        """every rank deposits its tensor; returns the list of all ranks' tensors (valid until the next exchange)"""
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.slots[rank] = t
        self.barrier.wait()
        got = list(self.slots)
        return got

    def done(self):
        self.barrier.wait()                                            # nobody overwrites its slot before everyone has read


class EmulatedShard:
    """drop-in for mcx.parallel.Shard (same methods), bound to (rank, world) of an EmulatedWorld"""
    active, backend, group = True, "nccl", None
    device_collectives, _trivial = True, False

    def __init__(self, rank: int, shared: EmulatedWorld):
        self.rank, self.world, self.shared = rank, shared.world, shared

    def split(self, n_total: int):
        base, rem = divmod(int(n_total), self.world)
        return self.rank * base + min(self.rank, rem), base + (1 if self.rank < rem else 0)

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        got = self.shared.exchange(self.rank, t)
        total = torch.stack([g.to(t.device) for g in got]).sum(dim=0)     # rank order: the same sum on every rank
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.shared.done()
        t.copy_(total)
        if self.rank == 0:
            self.shared.calls["all_reduce"] += 1
        return t

    def all_reduce_np(self, a: np.ndarray) -> np.ndarray:
        t = torch.from_numpy(np.ascontiguousarray(a).copy())
        return self.all_reduce_(t).numpy()

    def all_gather_dev(self, t: torch.Tensor) -> torch.Tensor:
        got = self.shared.exchange(self.rank, t.contiguous())
        out = torch.stack([g.to(t.device) for g in got])
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.shared.done()
        if self.rank == 0:
            self.shared.calls["all_gather"] += 1
        return out

    def all_gather_into(self, out: torch.Tensor, t: torch.Tensor) -> None:
        out.copy_(self.all_gather_dev(t))

    def all_gather_np(self, a: np.ndarray) -> np.ndarray:
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        return self.all_gather_dev(t).numpy()


def run_ranks(world: int, make_controller, body):
    """run body(controller, rank) on `world` emulated ranks (threads); returns the list of results in rank order.
    make_controller(rank) builds the rank's controller on its own backend."""
    shared = EmulatedWorld(world)
    out, err = [None] * world, [None] * world

    def work(rank):
        try:
            sc = make_controller(rank)
            sc.shard_factory = lambda: EmulatedShard(rank, shared)
            out[rank] = body(sc, rank)
        except BaseException as e:                                      # noqa: BLE001 — reported by the caller
            err[rank] = e
            shared.barrier.abort()

    threads = [threading.Thread(target=work, args=(r,), name=f"rank{r}") for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out, shared.calls
