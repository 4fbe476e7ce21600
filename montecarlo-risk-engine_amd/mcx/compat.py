"""Drop-in aliases: expose the mcx modules under the reference's top-level import paths
(`controller.controller`, `models.vasicek`, `metrics.cva_metric`, `products.swap`, `engine.engine`, `common.enums`, ...),
so scripts written against konstantineder/montecarlo-risk-engine (which put `src/` on sys.path) run unmodified.

    import mcx.compat; mcx.compat.install()
    from controller.controller import SimulationController      # now the MI355X-native one
"""
from __future__ import annotations

import importlib
import sys

_MODULES = [
    "common.packages", "common.enums",
    "controller.controller", "controller.simulation_results",
    "engine.engine",
    "helpers.cs_helper",
    "maths.maths", "maths.regression",
    "metrics.metric", "metrics.risk_metrics", "metrics.pv_metric", "metrics.ce_metric", "metrics.epe_metric",
    "metrics.ene_metric", "metrics.eepe_metric", "metrics.pfe_metric", "metrics.cva_metric",
    "models.model", "models.model_config", "models.black_scholes", "models.heston", "models.vasicek", "models.cirpp",
    "models.hull_white", "models.black_scholes_multi", "models.schwartz_two_factor",
    "products.product", "products.equity", "products.bond", "products.swap", "products.european_option",
    "products.bermudan_option", "products.netting_set", "products.basket_option", "products.binary_option", "products.asian_option", "products.barrier_option", "products.flexicall",
    "request_interface.request_types", "request_interface.request_interface",
]


def install() -> None:
    for pkg in sorted({m.split(".")[0] for m in _MODULES}):
        sys.modules.setdefault(pkg, importlib.import_module(f"mcx.{pkg}"))
    for m in _MODULES:
        sys.modules.setdefault(m, importlib.import_module(f"mcx.{m}"))
