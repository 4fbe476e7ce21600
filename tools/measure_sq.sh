#!/bin/bash
# Instruction mix / issue statistics of the bench kernels from the SQ counters (one --pmc pass per group, kernel-trace only).
# usage (on the GPU box): tools/measure_sq.sh <outdir> [bench args]
set -e
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  G=$((G+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_$G -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_sq_$G.json 2> $OUT/bench_sq_$G.err || echo "group $G failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/sq_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in acc.items():
    if not any(s in k for s in ("k1_paths", "kf_fused", "kf_lean")): continue
    row = {}
    for c, v in d.items():
        big = [x for x in v if x >= 0.5 * max(v)] if max(v) > 0 else v
        row[c] = sum(big) / len(big)
    res[k[:90]] = row
json.dump(res, open(f"{out}/sq_summary.json", "w"), indent=1)
# compact form read by bench.py (profiles/sq_counters.json)
compact = {}
for k, row in res.items():
    short = "k1_paths" if "k1_paths" in k else "kf_lean" if "kf_lean" in k else "kf_fused"
    if "SQ_WAVES" in row and "SQ_INSTS_VALU" in row:
        w = row["SQ_WAVES"]
        compact[short] = {"kernel": k, "waves": w, "n_simd": 1024, "valu_insts_per_wave": row["SQ_INSTS_VALU"] / w,
                          "salu_insts_per_wave": row.get("SQ_INSTS_SALU", 0) / w, "smem_insts_per_wave": row.get("SQ_INSTS_SMEM", 0) / w,
                          "lds_insts_per_wave": row.get("SQ_INSTS_LDS", 0) / w, "wave_cycles_per_wave": row.get("SQ_WAVE_CYCLES", 0) / w,
                          "wait_inst_any_per_wave": row.get("SQ_WAIT_INST_ANY", 0) / w,
                          "grbm_gui_active_sum_over_xcds": row.get("GRBM_GUI_ACTIVE", 0)}
json.dump(compact, open(f"{out}/sq_counters.json", "w"), indent=1)
for k, row in res.items():
    w = row.get("SQ_WAVES", 0) or 1
    print(k)
    print("  per wave:", {c: round(v / w, 1) for c, v in row.items() if c != "SQ_WAVES"}, "waves", w)
PY
