#!/bin/bash
O=$PWD/gpurun_out/r3p; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
for plan in semi unfused; do
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$plan -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-strong --sustain 0 --plan $plan --steps 100 > $O/bench_$plan.json 2> $O/prof_$plan.err )
f=$(find $O/prof_$plan -name "*kernel_stats.csv" | head -1); cp $f $O/bench_kernel_stats_$plan.csv
python3 - $O/bench_$plan.json $f $plan <<'PY'
import json,sys,csv
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[3], "ms/step %.4f"%d["ms_per_step"])
for r in list(csv.DictReader(open(sys.argv[2])))[:3]: print('   %-70s calls %5s avg_us %9.1f'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
timeout -k 10 600 python tools/run_configs.py 2 3 4 5 6 > $O/configs.jsonl 2> $O/configs.err; cut -c1-330 $O/configs.jsonl
timeout -k 10 300 python tools/large_book.py > $O/large_book.jsonl 2> $O/large_book.err; cut -c1-700 $O/large_book.jsonl
