#!/bin/bash
# A/B of the streaming passes (semi / unfused plans) under rocprofv3: variants/libmcx_<name>.so vs the in-tree library
O=$PWD/gpurun_out/${1:-abs}; mkdir -p $O; shift
for n in cur "$@"; do
  if [ $n == cur ]; then unset MCX_LIB_PATH; else export MCX_LIB_PATH=$GRAFT_REPO_ROOT/variants/libmcx_$n.so; fi
  for plan in ${PLANS:-semi unfused}; do
    ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${n}_$plan -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --plan $plan --steps 20 > $O/bench_${n}_$plan.json 2> $O/bench_${n}_$plan.err ) || { echo "$n $plan failed"; tail -3 $O/bench_${n}_$plan.err; }
    f=$(find $O/prof_${n}_$plan -name "*kernel_stats.csv" | head -1)
    echo "== $n $plan"; python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    print('   %-90s calls %5s avg_us %9.1f' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
  done
done
