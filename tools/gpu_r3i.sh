#!/bin/bash
O=$PWD/gpurun_out/r3i; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_full_size_configs.py tests/test_emulated_ranks.py -m gpu -x -q -k "select or config5 or emulated" > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/run_configs.py 5 > $O/cfg.jsonl 2> $O/cfg.err; cut -c1-500 $O/cfg.jsonl
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/tools/prof_cfg5.py > $O/prof.json 2> $O/prof.err ); echo "prof rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/config5_kernel_stats.csv; python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]: print('%-70s calls %5s avg_us %9.1f tot_ms %8.2f'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
