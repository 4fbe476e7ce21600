"""Host-side planning on ONE intra-op thread.

Everything this package does with torch on the host is small — timelines, 4 x 4 Cholesky factors, deep copies of model objects —
but torch sizes its intra-op pool from the core count of the machine (128 threads on the GPU hosts this was measured on), not from
the CPU quota of the container (16 CPUs).  The pool's threads spin between parallel regions; with a region every few milliseconds
the process burns its CFS quota in a fraction of each 100 ms period and is throttled for the rest of it: 13.5 s of CPU time and
8 throttled periods inside a 0.8 s bump-and-revalue run whose work is 0.115 s (tools/prof_bump.py reads cpu.stat before and after).
These were the "20-80 ms host stalls" of rounds 1-2.  The controller's entry points therefore run with the pool at one thread and
put the caller's setting back when they return."""
from __future__ import annotations

import functools

import torch


class single_threaded_host:
    """context manager / decorator: torch intra-op threads = 1 inside, restored on exit (re-entrant: a nested use is a no-op)"""

    def __enter__(self):
        self._n = torch.get_num_threads()
        if self._n > 1:
            torch.set_num_threads(1)
        return self

    def __exit__(self, *exc):
        if self._n > 1:
            torch.set_num_threads(self._n)
        return False

    def __call__(self, fn):
        @functools.wraps(fn)
        def wrapped(*a, **k):
            with single_threaded_host():
                return fn(*a, **k)
        return wrapped
