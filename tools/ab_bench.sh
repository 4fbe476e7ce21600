#!/bin/bash
# interleaved A/B timing of library variants (variants/libmcx_<name>.so, "cur" = the in-tree library) on the bench workload
# usage (GPU box): tools/ab_bench.sh <outdir> <rounds> name1 name2 ... [-- extra bench args]
OUT=$1; R=$2; shift 2; mkdir -p $OUT
NAMES=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do NAMES+=($1); shift; done; [ "$1" == "--" ] && shift
for r in $(seq 1 $R); do for n in "${NAMES[@]}"; do
  if [ $n == cur ]; then unset MCX_LIB_PATH; else export MCX_LIB_PATH=$PWD/variants/libmcx_$n.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --plan fused --steps 30 "$@" > $OUT/${n}_$r.json 2> $OUT/${n}_$r.err || { echo "$n failed"; tail -3 $OUT/${n}_$r.err; }
done; done
python - $OUT <<'PY'
import json,glob,sys,collections
res=collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); n=f.split('/')[-1].rsplit('_',1)[0]
        res[n].append((d['roofline']['kernel_ms'], d['ms_per_step'], d['result']['cva']))
    except Exception as e: print(f,'ERR',e)
for n,v in sorted(res.items(), key=lambda kv: min(x[0] for x in kv[1])):
    print('%-12s kernel_ms min %.4f  all %s  ms/step %.4f  cva %.15g'%(n, min(x[0] for x in v), ['%.4f'%x[0] for x in v], min(x[1] for x in v), v[0][2]))
PY
