"""Credit-curve helpers (reference: helpers/cs_helper.py:80-107 — only the piecewise-constant default probability is on
the hot path; the CDS bootstrap is out of scope)."""
from __future__ import annotations

import math


class CSHelper:
    def probability_of_default(self, hazards, tenors, date) -> float:
        """hazards[i] applies on (tenors[i-1], tenors[i]]; flat extension beyond the last tenor."""
        hazards = [float(h) for h in hazards]
        tenors = [float(t) for t in tenors]
        date = float(date)
        survival, prev = 1.0, 0.0
        idx = 0
        for idx, mat in enumerate(tenors):
            if mat <= date:
                survival *= math.exp(-hazards[idx] * (mat - prev))
                prev = mat
            else:
                break
        else:
            idx = len(tenors) - 1
        stub = date - prev
        if stub > 0:
            survival *= math.exp(-hazards[idx] * stub)
        return 1.0 - survival
