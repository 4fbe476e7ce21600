#!/bin/bash
# round 3: full GPU suite + the mcx side of the seed study on the round-3 kernels
O=$PWD/gpurun_out/r3m; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/seed_study.py 100000 100000 64 > $O/seed_100k.json 2> $O/seed.err && cat $O/seed_100k.json
timeout -k 10 300 python tools/seed_study.py 1048576 131072 32 > $O/seed_1m.json 2>> $O/seed.err && cat $O/seed_1m.json
