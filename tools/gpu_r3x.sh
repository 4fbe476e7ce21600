#!/bin/bash
# the semi / unfused plans of config 3 at 2^20 paths under rocprofv3 (after the padded leading dimension of the path / exposure /
# cashflow matrices): per-kernel device time of K1, the streaming date pass, K2 and K4
O=$PWD/gpurun_out/r3x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for plan in semi unfused; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$plan -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-strong --sustain 0 --plan $plan --steps 100 > $O/$plan.json 2> $O/$plan.err || { echo "$plan failed"; tail -3 $O/$plan.err; exit 1; }
  f=$(find $O/prof_$plan -name "*kernel_stats.csv" | head -1); cp $f $O/${plan}_kernel_stats.csv
  echo "== $plan"; grep "kf_lean\|k1_paths\|k2_eval\|k4_cva" $f | awk -F'",' '{split($2,a,","); printf("   %-70s avg %.1f us\n", substr($1,2,70), a[3]/1000)}'
  python3 -c "import json;d=json.load(open('$O/$plan.json'));print('   ms/step',d['ms_per_step'],'cva',d['result']['cva'])"
done
