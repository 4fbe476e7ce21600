"""MonteCarloEngine — the seam where HIP enters (reference: engine/engine.py:10-123).

Same constructor; `generate_paths()` returns a tensor indexed [path, date, state] exactly like the reference's
`torch.stack(paths, dim=1)`, but it is a VIEW of device memory laid out [date][state][path] (path contiguous, so the
64 lanes of a wavefront store/load 512 contiguous bytes).  The time loop, the correlated normals and every
simulate_time_step_* run inside one launch of the K1 kernel.

RNG: the reference seeds torch's global generator with 42 (pre-simulation) / 43 (main) in __init__ (engine.py:25); here
the same numbers are the Philox4x32-10 KEY, and the counter is (global path id, sub-step, draw), so results do not depend
on how paths are sharded over GPUs."""
from __future__ import annotations

import numpy as np
import torch

from .. import _native
from ..common.enums import SimulationScheme
from ..plan import SimPlan


class MonteCarloEngine:
    def __init__(self, simulation_timeline, simulation_type: SimulationScheme, model, num_paths: int, num_steps: int,
                 is_pre_simulation: bool = False, *, path_offset: int = 0, backend=None, plan: SimPlan | None = None,
                 sim=None, seed_offset: int = 0):
        self.simulation_type = simulation_type
        self.model = model
        self.num_paths = int(num_paths)
        self.num_steps = int(num_steps)
        self.simulation_timeline = simulation_timeline
        self.seed = (42 if is_pre_simulation else 43) + int(seed_offset)   # seed_offset: independent replications
        self.path_offset = int(path_offset)
        self.backend = backend if backend is not None else _native.get_backend()
        tl = simulation_timeline.detach().cpu().numpy() if isinstance(simulation_timeline, torch.Tensor) \
            else np.asarray(simulation_timeline, dtype=np.float64)
        self.plan = plan if plan is not None else SimPlan(model, tl, simulation_type, num_steps)
        self.sim = sim if sim is not None else self.backend.sim_create(self.plan)
        self.inject_z = None     # tests: replay recorded reference draws, [n_steps][n_z][N]
        self.inject_u = None

    def generate_paths_native(self, out=None) -> torch.Tensor:
        """[n_dates][n_state][num_paths] device tensor"""
        return self.backend.generate_paths(self.sim, self.seed, self.path_offset, self.num_paths,
                                           inject_z=self.inject_z, inject_u=self.inject_u, out=out)

    def generate_paths(self) -> torch.Tensor:
        return self.generate_paths_native().permute(2, 0, 1)
