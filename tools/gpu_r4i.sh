#!/bin/bash
O=$PWD/gpurun_out/r4i; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -30 $O/pytest.log | cut -c1-600; exit $rc
