#!/bin/bash
# round 3: tangent kernels without scratch — parity, configs 2 / 4 timings, kernel stats and SQ counters of config 4
O=$PWD/gpurun_out/${OUT_TAG:-r3k}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/run_configs.py 2 4 > $O/cfg.jsonl 2> $O/cfg.err; cut -c1-700 $O/cfg.jsonl
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/tools/run_configs.py 2 4 > $O/prof.json 2> $O/prof.err ); echo "prof rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/config24_kernel_stats.csv; python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]: print('%-80s calls %5s avg_us %9.1f'%(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3))
PY
cd /tmp && export TMPDIR=/tmp
G=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  G=$((G+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq_$G -- python3 $GRAFT_REPO_ROOT/tools/run_configs.py 4 > $O/sq_$G.json 2> $O/sq_$G.err || echo "group $G failed"
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
    if not any(s in k for s in ("kt_heston", "kf_lean")): continue
    row = {}
    for c, v in d.items():
        big = [x for x in v if x >= 0.5 * max(v)] if max(v) > 0 else v
        row[c] = sum(big) / len(big)
    res[k[:100]] = row
json.dump(res, open(f"{out}/config4_sq_summary.json", "w"), indent=1)
for k, row in res.items():
    w = row.get("SQ_WAVES", 0) or 1
    print(k); print("  per wave:", {c: round(v / w, 1) for c, v in row.items() if c != "SQ_WAVES"}, "waves", w)
PY
