#!/bin/bash
# streaming date pass A/B: library variants under variants/ against the in-tree library ("cur"), kernel time from rocprofv3
O=$PWD/gpurun_out/r3v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  if [ $n == cur ]; then unset MCX_LIB_PATH; else export MCX_LIB_PATH=$GRAFT_REPO_ROOT/variants/libmcx_$n.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-strong --sustain 0 --plan semi --steps 100 > $O/semi_$n.json 2> $O/semi_$n.err || { echo "$n failed"; tail -3 $O/semi_$n.err; exit 1; }
  f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1); cp $f $O/semi_${n}_kernel_stats.csv
  echo "== $n  $(grep kf_lean $O/semi_${n}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200)"
  python3 -c "import json;d=json.load(open('$O/semi_$n.json'));print('   ms/step',d['ms_per_step'],'cva',d['result']['cva'])"
done
