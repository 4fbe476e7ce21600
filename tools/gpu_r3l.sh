#!/bin/bash
O=$PWD/gpurun_out/r3l; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "heston or config4 or aad or tangent or golden or inject" > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/run_configs.py 4 > $O/cfg.jsonl 2> $O/cfg.err; cut -c1-700 $O/cfg.jsonl
