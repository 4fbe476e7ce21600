#!/bin/bash
# HBM traffic of the bench kernels from the PMC counters, collected as MI355X_MICROARCH.md prescribes:
# separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only; on gfx950 FETCH_SIZE reports
# exactly half of the bytes of a coalesced streaming read (confirmed here on K2: 2 x 428.6 MB = 857 MB = 2 columns x 51 dates
# x 8 B x 2^20 paths), WRITE_SIZE is exact (K1: 1,711,276,032 B = 51 x 4 x 8 B x 2^20).
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
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/pmc_{C}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        big = [x for x in v if x >= 0.5 * max(v)]          # main-simulation launches (the pre-simulation launch is 8x smaller)
        res[k][C] = {"launches": len(v), "avg_main_launch_KB": sum(big) / len(big)}
summary = {}
for k, v in res.items():
    short = "k1_paths" if "k1_paths" in k else "kf_lean" if "kf_lean" in k else "kf_fused" if "kf_fused" in k else "k2_eval_book" if "k2_eval" in k else \
            "k4_cva_paths" if "k4_cva" in k else None
    if short:
        fetch = v.get("FETCH_SIZE", {}).get("avg_main_launch_KB", 0.0) * 1024 * 2      # gfx950 correction
        write = v.get("WRITE_SIZE", {}).get("avg_main_launch_KB", 0.0) * 1024
        summary[short] = {"kernel": k, "fetch_bytes_corrected": fetch, "write_bytes": write,
                          "hbm_bytes_per_launch_at_1Mi_paths": fetch + write}
json.dump({"raw_KB": res, "summary": summary}, open(f"{out}/pmc_summary.json", "w"), indent=1)
for k, v in summary.items():
    print(k, {a: round(b) for a, b in v.items() if a != "kernel"})
PY
