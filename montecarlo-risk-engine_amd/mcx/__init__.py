"""mcx — MI355X-native Monte-Carlo path-generation and exposure engine.

Drop-in for the hot path of konstantineder/montecarlo-risk-engine: same class surface (SimulationController,
ModelConfig + models, products, RiskMetrics + metrics, SimulationResults), the per-timestep SDE evolution and the
payoff / exposure reductions run as hand-written HIP kernels for gfx950 behind the C ABI of include/mcx.h."""
from .common.enums import SimulationScheme  # noqa: F401

__version__ = "0.1.0"
