#!/bin/bash
# shader clock held during the launches of tools/ubench_valu (or any command): GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs)
# / kernel duration.  usage (GPU box): tools/measure_clock.sh <outdir> [command ...]   (default command: tools/ubench_valu)
set -e
OUT=$1; shift
mkdir -p $OUT
CMD=${@:-$GRAFT_REPO_ROOT/tools/ubench_valu}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/clk -- $CMD > $OUT/clk_stdout.txt 2> $OUT/clk.err
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
dur = {}
for f in glob.glob(f"{out}/clk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
rows = collections.defaultdict(list)
for f in glob.glob(f"{out}/clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            name, ns = dur[r["Dispatch_Id"]]
            if ns > 5e6:                                   # the long launches only
                rows[name[:60]].append(float(r["Counter_Value"]) / 8.0 / ns * 1e3)
with open(f"{out}/clock_mhz.txt", "w") as fo:
    for k, v in rows.items():
        line = "%-62s %4d launches  shader clock %.0f MHz (min %.0f, max %.0f)" % (k, len(v), sum(v) / len(v), min(v), max(v))
        print(line); fo.write(line + "\n")
PY
