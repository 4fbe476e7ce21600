#!/bin/bash
O=$PWD/gpurun_out/r3f; mkdir -p $O
for p in 131072 262144; do
  for cfg in "11 0" "11 21000" "11 28000" "21 0" "21 21000" "21 28000"; do
    set -- $cfg
    MCX_LIB_PATH=$PWD/variants/libmcx_ab.so MCX_LEAN_SHAPE=$1 MCX_LEAN_LDS_PAD=$2 timeout -k 10 120 python bench.py --paths $p --no-cpu-baseline --no-strong --sustain 0 --plan fused --steps 40 > $O/b_${p}_$1_$2.json 2> $O/b_${p}_$1_$2.err || { tail -3 $O/b_${p}_$1_$2.err; continue; }
    python3 - $O/b_${p}_$1_$2.json $p "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("paths %s shape/pad %s  ms/step %.4f kernel_ms %.4f cva %.10f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["roofline"]["kernel_ms"], d["result"]["cva"]), flush=True)
PY
  done
done
