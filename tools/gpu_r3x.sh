#!/bin/bash
# padded leading dimension of the paths tensor: GPU suite, then the semi / unfused plans at 2^20 paths (four paths per lane in the
# streaming pass = in-tree, two = variants/libmcx_sp2.so)
O=$PWD/gpurun_out/r3x; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
for n in cur; do for plan in unfused; do
  if [ $n == cur ]; then unset MCX_LIB_PATH; else export MCX_LIB_PATH=$GRAFT_REPO_ROOT/variants/libmcx_$n.so; fi
  [ $n == sp2 ] && [ $plan == unfused ] && continue
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${plan}_$n -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-strong --sustain 0 --plan $plan --steps 100 > $O/${plan}_$n.json 2> $O/${plan}_$n.err || { echo "$n failed"; tail -3 $O/${plan}_$n.err; exit 1; }
  f=$(find $O/prof_${plan}_$n -name "*kernel_stats.csv" | head -1); cp $f $O/${plan}_${n}_kernel_stats.csv
  echo "== $plan $n"; grep "kf_lean\|k1_paths\|k2_eval\|k4_cva" $f | awk -F'",' '{split($2,a,","); printf("   %-70s avg %.1f us\n", substr($1,2,70), a[3]/1000)}'
  python3 -c "import json;d=json.load(open('$O/${plan}_$n.json'));print('   ms/step',d['ms_per_step'],'cva',d['result']['cva'])"
done; done
cd $GRAFT_REPO_ROOT
for b in cva ee pv; do
  timeout -k 10 300 python tools/large_book.py --book $b --repeat 3 > $O/large_book_$b.json 2> $O/large_book_$b.err || { tail -5 $O/large_book_$b.err; exit 1; }
  python3 -c "
import json
for l in open('$O/large_book_$b.json'):
    d=json.loads(l); print('$b run_s %.3f  products/s %.0f  %s %s'%(d['run_s'], d['products_per_second'], d['prepare'], {k:d[k] for k in ('cva','pv','peak_epe','peak_pfe') if k in d}))"
done
