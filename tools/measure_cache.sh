#!/bin/bash
# scalar-data-cache and instruction-cache behaviour of the bench kernels (SQC counters, kernel-trace only)
set -e
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G=0
for C in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_TC_STALL" "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES"; do
  G=$((G+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/c_$G -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_c_$G.json 2> $OUT/bench_c_$G.err || echo "group $G failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/c_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in acc.items():
    if not any(s in k for s in ("k1_paths", "kf_fused")): continue
    res[k[:80]] = {c: (sum(x for x in v if x >= 0.5 * max(v)) / max(1, sum(1 for x in v if x >= 0.5 * max(v)))) for c, v in d.items()}
json.dump(res, open(f"{out}/cache_summary.json", "w"), indent=1)
for k, row in res.items():
    print(k); print("  ", {c: round(v) for c, v in row.items()})
PY
