#!/bin/bash
# config 5 (2 M paths x 120 dates): HBM bytes fetched / written by every kernel of one run (FETCH_SIZE x 2 on gfx950, WRITE_SIZE; separate
# --pmc passes): how often is the 2.03 GB exposure matrix read by the PFE select?
O=$PWD/gpurun_out/r3traffic5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- python3 $GRAFT_REPO_ROOT/tools/prof_cfg5.py > $O/run_$C.json 2> $O/run_$C.err || { tail -3 $O/run_$C.err; exit 1; }
done
python3 - $O <<'PY'
import csv, glob, json, sys, collections, re
out = sys.argv[1]
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^>]*>)?)", n)
    return (m.group(1) if m else n)[:48]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/pmc_{C}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C:
                k = short(r["Kernel_Name"])
                tot[k][C] += float(r["Counter_Value"]) * 1024 * (2 if C == "FETCH_SIZE" else 1)
                if C == "FETCH_SIZE": cnt[k] += 1
runs = 3.0                                                      # tools/prof_cfg5.py runs the configuration three times
matrix = 8.0 * 121 * (1 << 21)
res = {k: {"launches_per_run": cnt[k] / runs, "fetched_GB_per_run": v["FETCH_SIZE"] / runs / 1e9, "written_GB_per_run": v["WRITE_SIZE"] / runs / 1e9} for k, v in tot.items()}
sel = sum(v["fetched_GB_per_run"] for k, v in res.items() if k.startswith("k5_"))
summary = {"exposure_matrix_GB": matrix / 1e9, "select_kernels_fetched_GB_per_run": sel, "select_reads_of_the_matrix": sel / (matrix / 1e9), "kernels": res}
json.dump(summary, open(f"{out}/config5_traffic.json", "w"), indent=1)
print("matrix %.3f GB; select kernels fetch %.3f GB per run = %.2f x the matrix" % (matrix / 1e9, sel, sel / (matrix / 1e9)))
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["fetched_GB_per_run"])[:8]: print("  %-40s %6.1f launches  fetched %.3f GB  written %.3f GB" % (k, v["launches_per_run"], v["fetched_GB_per_run"], v["written_GB_per_run"]))
PY
