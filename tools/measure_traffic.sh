#!/bin/bash
# HBM traffic of the bench kernels from the PMC counters, collected as MI355X_MICROARCH.md prescribes:
# separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only.
# usage (on the GPU box): tools/measure_traffic.sh <outdir> [bench args]
set -e
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_$C.json 2> $OUT/bench_$C.err
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{out}/pmc_{C}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        big = [x for x in v if x >= 0.5 * max(v)]          # main-simulation launches (the pre-simulation launch is 8x smaller)
        res[k][C] = {"launches": len(v), "avg_main_launch": sum(big) / len(big), "max": max(v)}
json.dump(res, open(f"{out}/pmc_summary.json", "w"), indent=1)
for k, v in res.items():
    print(k[:70], {c: round(x["avg_main_launch"]) for c, x in v.items()})
PY
