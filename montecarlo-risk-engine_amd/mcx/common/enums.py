"""Simulation schemes; values mirror the reference's common/enums.py:4-9 and MCX_SCHEME_* in include/mcx.h."""
from __future__ import annotations

from enum import Enum


class SimulationScheme(Enum):
    EULER = 0
    MILSTEIN = 1      # declared by the reference but never implemented there (models/model.py:129-133)
    ANALYTICAL = 2
    QE = 3
