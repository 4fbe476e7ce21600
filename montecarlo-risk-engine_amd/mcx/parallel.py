"""Path sharding across the GPUs of one node: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm)
or gloo (CPU tests).  Paths are independent, so no path data ever crosses GPUs; only accumulator records, LSM normal
equations and radix-select histograms are exchanged."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


class Shard:
    """global path range of this rank: contiguous, remainder spread over the first ranks"""

    def __init__(self, group=None):
        self.active = dist.is_available() and dist.is_initialized()
        self.group = group
        self.rank = dist.get_rank(group) if self.active else 0
        self.world = dist.get_world_size(group) if self.active else 1
        self.backend = dist.get_backend(group) if self.active else None

    def split(self, n_total: int) -> tuple[int, int]:
        base, rem = divmod(int(n_total), self.world)
        count = base + (1 if self.rank < rem else 0)
        offset = self.rank * base + min(self.rank, rem)
        return offset, count

    def _comm_device(self, like: torch.Tensor | None = None):
        if self.backend == "nccl":
            return like.device if like is not None and like.is_cuda else torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")

    @property
    def _trivial(self) -> bool:
        """no collective needed: no process group, or a one-rank gloo group.  A one-rank RCCL group still goes through the
        collectives, so that a single GPU exercises exactly the code eight GPUs run"""
        return not self.active or (self.world == 1 and self.backend != "nccl")

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        """in-place sum over ranks of a device (nccl) or host (gloo) tensor"""
        if self._trivial:
            return t
        dev = self._comm_device(t)
        if t.device == dev:
            dist.all_reduce(t, group=self.group)
            return t
        tmp = t.to(dev)
        dist.all_reduce(tmp, group=self.group)
        t.copy_(tmp)
        return t

    def all_reduce_np(self, a: np.ndarray) -> np.ndarray:
        if self._trivial:
            return a
        t = torch.from_numpy(np.ascontiguousarray(a)).to(self._comm_device())
        dist.all_reduce(t, group=self.group)
        return t.cpu().numpy()

    @property
    def device_collectives(self) -> bool:
        """True when the process group moves device memory (RCCL): accumulator records then stay on the GPU until gathered
        (also for a one-rank group, so that a single GPU exercises the same code path as eight)"""
        return self.active and self.backend == "nccl"

    def all_gather_dev(self, t: torch.Tensor) -> torch.Tensor:
        """[world, *t.shape] on the device of t: ONE collective over xGMI, no host hop (the caller copies the result to the
        host once per pass)"""
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out

    def all_gather_into(self, out: torch.Tensor, t: torch.Tensor) -> None:
        """the same collective into a caller-owned [world, *t.shape] device buffer, on the CURRENT stream (the pipelined passes
        of the controller issue it on a side stream)"""
        dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)

    def all_gather_np(self, a: np.ndarray) -> np.ndarray:
        """stack equal-shaped float64 arrays of all ranks along a new leading axis (the single small collective that
        carries the (n, shift, s1, s2) accumulator records of every metric)"""
        a = np.ascontiguousarray(a, dtype=np.float64)
        if self._trivial:
            return a[None]
        t = torch.from_numpy(a).to(self._comm_device())
        if self.backend == "nccl":
            return self.all_gather_dev(t).cpu().numpy()
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return np.stack([o.numpy() for o in out], axis=0)
