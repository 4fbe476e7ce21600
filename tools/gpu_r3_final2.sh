#!/bin/bash
# round 3, final measurement set, part 2: the default bench line (CPU baseline on all cores), the other two execution plans under
# rocprofv3, BASELINE configs 2-5 + sensitivities, kernel stats of configs 2 / 4 / 5, the three large books, the bump path
O=$PWD/gpurun_out/r3final; mkdir -p $O
cp $O/counters/counters.json profiles/counters.json 2>/dev/null
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 - $O/bench_default.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("default: ms/step %.4f kernel_ms %.4f value %.4e frac %.4f alu %s cpu %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["value"], d["roofline"]["frac"], (d.get("alu") or {}).get("frac"), {k: d["cpu_baseline"][k] for k in ("value", "cores")}))
PY
for plan in semi unfused; do
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$plan -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-strong --sustain 0 --plan $plan --steps 100 > $O/bench_$plan.json 2> $O/prof_$plan.err )
f=$(find $O/prof_$plan -name "*kernel_stats.csv" | head -1); cp $f $O/bench_kernel_stats_$plan.csv
python3 - $O/bench_$plan.json $f $plan <<'PY'
import json,sys,csv
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[3], "ms/step %.4f"%d["ms_per_step"])
for r in list(csv.DictReader(open(sys.argv[2])))[:3]: print('   %-70s calls %5s avg_us %9.1f'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
timeout -k 10 600 python tools/run_configs.py 2 3 4 5 6 > $O/configs.jsonl 2> $O/configs.err; cut -c1-300 $O/configs.jsonl
timeout -k 10 300 python tools/run_configs.py 7 > $O/config5_sensitivities.jsonl 2> $O/config5_sensitivities.err; cut -c1-300 $O/config5_sensitivities.jsonl
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 $GRAFT_REPO_ROOT/tools/prof_cfg5.py > $O/prof_c5.json 2> $O/prof_c5.err )
f=$(find $O/prof_c5 -name "*kernel_stats.csv" | head -1); cp $f $O/config5_kernel_stats.csv; python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]: print('%-70s calls %5s avg_us %9.1f tot_ms %8.2f'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c24 -- python3 $GRAFT_REPO_ROOT/tools/run_configs.py 2 4 > $O/prof_c24.json 2> $O/prof_c24.err )
f=$(find $O/prof_c24 -name "*kernel_stats.csv" | head -1); cp $f $O/config2_config4_kernel_stats.csv; python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]: print('%-70s calls %5s avg_us %9.1f'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
for b in cva ee pv; do
  timeout -k 10 300 python tools/large_book.py --book $b --repeat 3 > $O/large_book_$b.json 2> $O/large_book_$b.err || { tail -5 $O/large_book_$b.err; exit 1; }
  python3 -c "
import json
for l in open('$O/large_book_$b.json'):
    d=json.loads(l); print('$b run_s %.3f  products/s %.0f  %s'%(d['run_s'], d['products_per_second'], d['prepare']))"
done
timeout -k 10 300 python tools/prof_bump.py default > $O/prof_bump.txt 2> $O/prof_bump.err; grep "bumps\|process CPU" $O/prof_bump.txt | cut -c1-200
timeout -k 10 300 python tools/prof_bump.py unguarded > $O/prof_bump_unguarded.txt 2> $O/prof_bump_unguarded.err; grep "bumps\|process CPU" $O/prof_bump_unguarded.txt | cut -c1-200
