#!/bin/bash
O=$PWD/gpurun_out/r3n; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "forward_mode or lsm_sensitivities or tangent" > $O/pytest.log 2>&1; rc=$?; tail -25 $O/pytest.log; exit $rc
