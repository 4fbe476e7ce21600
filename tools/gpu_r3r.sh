#!/bin/bash
# SQ counters of config 5's kernels
O=$PWD/gpurun_out/r3r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  G=$((G+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq_$G -- python3 $GRAFT_REPO_ROOT/tools/prof_cfg5.py > $O/sq_$G.json 2> $O/sq_$G.err || echo "group $G failed"
done
python3 - $O <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/sq_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in acc.items():
    if not any(s in k for s in ("kf_lean", "k3_step", "k5_bracket")): continue
    row = {}
    for c, v in d.items():
        big = [x for x in v if x >= 0.5 * max(v)] if max(v) > 0 else v
        row[c] = sum(big) / len(big)
    res[k[:100]] = row
json.dump(res, open(f"{out}/config5_sq_summary.json", "w"), indent=1)
for k, row in res.items():
    w = row.get("SQ_WAVES", 0) or 1
    print(k); print("  per wave:", {c: round(v / w, 1) for c, v in row.items() if c != "SQ_WAVES"}, "waves", w)
PY
