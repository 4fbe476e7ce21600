#!/bin/bash
# Counters of the dominant bench kernels for bench.py's `roofline.traffic` and `alu` objects: SQ instruction mix (tools/measure_sq.sh)
# and HBM bytes (tools/measure_traffic.sh), each in its own rocprofv3 --pmc passes, stamped with the hash of the kernel sources
# they were measured on (bench.py drops them when the library was built from other sources).
# usage (on the GPU box): tools/measure_counters.sh <outdir> [bench args]     -> <outdir>/counters.json (copy to profiles/)
set -e
OUT=$1; shift
HERE=$(cd $(dirname $0) && pwd)
mkdir -p $OUT
$HERE/ubench_valu > $OUT/ubench_valu.txt 2>&1 || { echo "tools/ubench_valu missing: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu"; exit 1; }
python3 $HERE/asm_mix.py > $OUT/asm_mix.json
$HERE/measure_sq.sh $OUT/sq "$@" > $OUT.sq.log 2>&1 || { tail -5 $OUT.sq.log; exit 1; }
$HERE/measure_traffic.sh $OUT/traffic "$@" > $OUT.traffic.log 2>&1 || { tail -5 $OUT.traffic.log; exit 1; }
python3 - "$OUT" "$GRAFT_REPO_ROOT" <<'PY'
import json, sys, os
out, root = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
import bench
sq = json.load(open(f"{out}/sq/sq_counters.json"))
tr = json.load(open(f"{out}/traffic/pmc_summary.json"))["summary"]
res = {"kernel_source_sha": bench.kernel_source_sha(),
       "note": "rocprofv3 --pmc, separate passes (tools/measure_counters.sh), bench.py at 2^20 paths; per-wave averages over the main-simulation launches; FETCH_SIZE doubled (gfx950), WRITE_SIZE exact"}
# sustained issue cost (ns per wave64 instruction per SIMD, every CU busy, 8 waves per SIMD) of the instruction classes,
# measured by tools/ubench_valu in this very call: under f64 VALU load the chip does NOT hold 2.4 GHz
ub = {}
for line in open(f"{out}/ubench_valu.txt"):
    name = line[:36].strip()
    if "ns per wave-instr" in line:
        ub[name] = float(line.split(")")[-2].split("ns per wave-instr")[0].strip().split()[-1])
res["issue_ns"] = {"f64": ub["v_fma_f64"], "mad_u64_u32": ub["v_mad_u64_u32 (+shift)"], "int32": ub["xor/shift int (3 ops)"],
                   "rsq_f64": ub["v_rsq_f64"], "minmax_f64": ub["v_max_f64"]}
res["issue_ns_source"] = "tools/ubench_valu: HIP events around ONE >= 10 ms launch / wave-instructions per SIMD (8 independent chains per lane, 8 waves per SIMD)"
res["substep_mix_per_path"] = json.load(open(f"{out}/asm_mix.json"))["per_path_substep"]
for k, v in sq.items():
    e = dict(v)
    if k in tr:
        e.update({a: b for a, b in tr[k].items() if a != "kernel"})
    res[k] = e
json.dump(res, open(f"{out}/counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
