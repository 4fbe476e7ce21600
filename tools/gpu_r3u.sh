#!/bin/bash
# (1) streaming date pass: four paths per lane (variants/libmcx_ppl4.so) against the in-tree two; (2) kernel stats of the large book
O=$PWD/gpurun_out/r3u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in cur ppl4; do
  if [ $n == cur ]; then unset MCX_LIB_PATH; else export MCX_LIB_PATH=$GRAFT_REPO_ROOT/variants/libmcx_$n.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-strong --sustain 0 --plan semi --steps 100 > $O/semi_$n.json 2> $O/semi_$n.err || { echo "$n failed"; tail -3 $O/semi_$n.err; exit 1; }
  f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1); cp $f $O/semi_${n}_kernel_stats.csv
  echo "== $n"; head -4 $O/semi_${n}_kernel_stats.csv | cut -c1-220
  python3 -c "import json;d=json.load(open('$O/semi_$n.json'));print('ms/step',d['ms_per_step'],'cva',d['result']['cva'])"
done
unset MCX_LIB_PATH
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lb -- python3 $GRAFT_REPO_ROOT/tools/large_book.py --repeat 3 > $O/large_book.json 2> $O/large_book.err || { tail -5 $O/large_book.err; exit 1; }
f=$(find $O/prof_lb -name "*kernel_stats.csv" | head -1); cp $f $O/large_book_kernel_stats.csv; head -12 $O/large_book_kernel_stats.csv | cut -c1-200
python3 -c "
import json
for l in open('$O/large_book.json'):
    d=json.loads(l); print('run_s %.3f  products/s %.0f  %s'%(d['run_s'], d['products_per_second'], d['prepare']))"
