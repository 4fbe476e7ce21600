"""ModelConfig: several correlated sub-models simulated jointly (reference: models/model_config.py:8-307).

Keeps the reference's constructor, id routing (asset id / "numeraire" / "discount" -> sub-model), state concatenation,
block correlation / covariance assembly and parameter flattening; the joint Cholesky factor is what the path kernel
multiplies the Philox normals with."""
from __future__ import annotations

import numpy as np
import torch

from ..common.enums import SimulationScheme
from ..common.packages import FLOAT, device
from .black_scholes import BlackScholesModel
from .model import Model


class ModelConfig(Model):
    def __init__(self, models: list[Model], numeraire_model_idx: int = 0, discount_model_idx: int = 0,
                 inter_asset_correlation_matrix=None):
        assert len(models) > 0, "Provide at least one model."
        assert all(models[i].t0() == models[i + 1].t0() for i in range(len(models) - 1)), \
            "All models must share the same calibration_date."
        asset_ids = [a for m in models for a in m.asset_ids]
        assert len(asset_ids) == len(set(asset_ids)), \
            "Duplicate asset_ids detected across models. A particular asset can only be simulated by one distinct model."
        super().__init__(calibration_date=models[0].t0(), asset_ids=asset_ids,
                         simulation_dim=sum(m.simulation_dim for m in models),
                         state_dim=sum(m.state_dim for m in models))
        self.models = models
        self.id_to_model: dict = {"numeraire": numeraire_model_idx, "discount": discount_model_idx}
        for idx, m in enumerate(models):
            for a in m.asset_ids:
                self.id_to_model[a] = idx
        self.model_state_offset: dict[int, int] = {}
        off = 0
        for idx, m in enumerate(models):
            self.model_state_offset[idx] = off
            off += m.state_dim
        self.model_params = [p for m in models for p in m.get_model_params()]
        # the reference iterates the argument element-wise (model_config.py:67-78): a bare np.array([rho]) and a list
        # of 2-D blocks are both accepted, ordered over model pairs (i < j)
        self.inter_asset_correlation_matrix: list[torch.Tensor] = []
        if inter_asset_correlation_matrix is None:
            for i, m1 in enumerate(models):
                for m2 in models[i + 1:]:
                    self.inter_asset_correlation_matrix.append(torch.zeros(m1.num_assets, m2.num_assets, dtype=FLOAT))
        else:
            for block in inter_asset_correlation_matrix:
                self.inter_asset_correlation_matrix.append(torch.tensor(np.asarray(block), dtype=FLOAT, device=device))

    def requires_grad(self):
        self.perform_smoothing = True
        for m in self.models:
            m.requires_grad()

    def get_model_param_names(self) -> list[str]:
        names = []
        for m in self.models:
            label = m.asset_ids[0] if len(m.asset_ids) == 1 and m.asset_ids[0] else m.__class__.__name__
            names += [f"{label}.{n}" for n in m.get_model_param_names()]
        return names

    def _assemble(self, diag_block, off_block) -> torch.Tensor:
        n = self.num_assets
        out = torch.zeros((n, n), dtype=FLOAT, device=device)
        row, k = 0, 0
        for i, m1 in enumerate(self.models):
            n1 = m1.num_assets
            out[row:row + n1, row:row + n1] = diag_block(m1)
            col = row + n1
            for m2 in self.models[i + 1:]:
                n2 = m2.num_assets
                blk = off_block(m1, m2, self.inter_asset_correlation_matrix[k])
                out[row:row + n1, col:col + n2] = blk
                out[col:col + n2, row:row + n1] = blk.transpose(-1, -2) if blk.ndim >= 2 else blk
                col += n2
                k += 1
            row += n1
        return 0.5 * (out + out.T)

    def _get_correlation_matrix(self, simulation_scheme) -> torch.Tensor:
        return self._assemble(lambda m: m._get_correlation_matrix(simulation_scheme).detach(), lambda m1, m2, c: c)

    def _get_covariance_matrix(self, delta_t) -> torch.Tensor:
        dt = float(delta_t)

        def inter(m1, m2, corr):
            if isinstance(m1, BlackScholesModel) and isinstance(m2, BlackScholesModel):
                return torch.outer(m1.get_volatility().detach(), m2.get_volatility().detach()) * corr * dt
            raise NotImplementedError("Inter covariance not implemented for the requested pair of models.")

        return self._assemble(lambda m: m._get_covariance_matrix(dt).detach(), inter)

    # ---- native hooks -------------------------------------------------------------------------------------------
    def _slots(self):
        return [s for m in self.models for s in m._slots()]

    def _initial_state(self):
        return [v for m in self.models for v in m._initial_state()]

    def _step_aux(self, scheme, t1, dt):
        return [a for m in self.models for a in m._step_aux(scheme, t1, dt)]

    def _n_uniform(self, scheme):
        return 0

    def _supports_scheme(self, scheme):
        if scheme == SimulationScheme.QE:
            return False      # model_config.py has no simulate_time_step_qe
        return all(m._supports_scheme(scheme) for m in self.models)

    def _route(self, asset_id):
        idx = self.id_to_model[asset_id]
        return self.models[idx], self.model_state_offset[idx]

    def _atom(self, req, asset_id):
        model, off = self._route(asset_id)
        at = model._atom(req, asset_id)
        if at.col is not None:
            at.col += off
        return at

    def resolve_request(self, req, asset_id, state):
        return self._atom(req, asset_id).evaluate(state)
