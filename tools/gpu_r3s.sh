#!/bin/bash
# round 3, second session: full GPU suite, default bench, large book (one-call batched LSM, word-copied event blocks)
O=$PWD/gpurun_out/r3s; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
python3 - $O/bench_default.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("default: ms/step %.4f kernel_ms %.4f value %.4e frac %.4f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["value"], d["roofline"]["frac"]))
PY
timeout -k 10 300 python tools/large_book.py --repeat 4 --profile > $O/large_book.json 2> $O/large_book.err || exit 1
cat $O/large_book.json | cut -c1-700
grep -A 40 'Ordered by' $O/large_book.err | cut -c1-160
