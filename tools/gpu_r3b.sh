#!/bin/bash
# round 3: value polynomials — parity tests, config 5 timing and kernel stats
O=$PWD/gpurun_out/r3b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_value_polynomials.py tests/test_hip_parity.py tests/test_full_size_configs.py tests/test_edge_cases.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/run_configs.py 5 3 > $O/cfg.jsonl 2> $O/cfg.err; cut -c1-600 $O/cfg.jsonl
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/tools/prof_cfg5.py > $O/prof.json 2> $O/prof.err ); echo "prof rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/config5_kernel_stats.csv; python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]: print('%-70s calls %5s avg_us %9.1f'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
