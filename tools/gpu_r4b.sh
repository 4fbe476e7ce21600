#!/bin/bash
O=$PWD/gpurun_out/r3final; mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 - $O/bench_default.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("default: ms/step %.4f kernel_ms %.4f value %.4e frac %.4f alu %s traffic %s cpu %s sustained %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["value"], d["roofline"]["frac"], (d.get("alu") or {}).get("frac"), d["roofline"]["traffic"], {k: d["cpu_baseline"][k] for k in ("value", "cores")}, d["sustained"]))
PY
