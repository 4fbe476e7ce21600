#!/bin/bash
# round 3: A/B of the kf_lean launch shapes at small path counts (variants/libmcx_ab.so, MCX_LEAN_SHAPE = 10 * PPL + RU)
O=$PWD/gpurun_out/r3c; mkdir -p $O
for p in 131072 262144 524288; do
  for sh in 21 11 12 14 15 22 24; do
    MCX_LIB_PATH=$PWD/variants/libmcx_ab.so MCX_LEAN_SHAPE=$sh timeout -k 10 120 python bench.py --paths $p --no-cpu-baseline --plan fused --steps 40 > $O/b_${p}_$sh.json 2> $O/b_${p}_$sh.err || exit 1
    python3 - $O/b_${p}_$sh.json $p $sh <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("paths %s shape %s  ms/step %.4f kernel_ms %.4f cva %.10f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["roofline"]["kernel_ms"], d["result"]["cva"]), flush=True)
PY
  done
done
# product library: default heuristic
for p in 131072 262144 524288 1048576; do
  timeout -k 10 120 python bench.py --paths $p --no-cpu-baseline --plan fused --steps 40 > $O/prod_$p.json 2> $O/prod_$p.err || exit 1
  python3 - $O/prod_$p.json $p prod <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("paths %s shape %s  ms/step %.4f kernel_ms %.4f cva %.10f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["roofline"]["kernel_ms"], d["result"]["cva"]), flush=True)
PY
done
