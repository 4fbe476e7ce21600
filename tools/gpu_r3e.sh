#!/bin/bash
# A/B of launch shapes (variants/libmcx_ab.so) + RNG parity tests
O=$PWD/gpurun_out/r3e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_full_size_configs.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
for p in 131072 262144 1048576; do
  for sh in ${SHAPES:-21 11}; do
    MCX_LIB_PATH=$PWD/variants/libmcx_ab.so MCX_LEAN_SHAPE=$sh timeout -k 10 120 python bench.py --paths $p --no-cpu-baseline --no-strong --sustain 0 --plan fused --steps 40 > $O/b_${p}_$sh.json 2> $O/b_${p}_$sh.err || exit 1
    python3 - $O/b_${p}_$sh.json $p $sh <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("paths %s shape %s  ms/step %.4f kernel_ms %.4f cva %.10f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["roofline"]["kernel_ms"], d["result"]["cva"]), flush=True)
PY
  done
done
