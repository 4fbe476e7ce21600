#!/bin/bash
# HBM traffic (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate --pmc passes) of the streaming kernels of config 3's semi and unfused
# plans at 2^20 paths, next to their algorithmic bytes: traffic well above them would mean re-reads
O=$PWD/gpurun_out/r3traffic; mkdir -p $O
for plan in semi unfused; do
  timeout -k 10 400 bash tools/measure_traffic.sh $O/$plan --plan $plan --no-strong --sustain 0 > $O/$plan.log 2>&1 || { tail -5 $O/$plan.log; exit 1; }
  echo "== $plan"; tail -4 $O/$plan.log | cut -c1-240
done
python3 - $O <<'PY'
import json, sys
out = sys.argv[1]
N, T, D, E = 1 << 20, 51, 4, 51
alg = {"k1_paths": 8 * T * D * N, "kf_lean": 8 * T * D * N, "k2_eval_book": 8 * (2 * T + E) * N, "k4_cva_paths": 8 * (E + 2 * (E - 1)) * N}
res = {}
for plan in ("semi", "unfused"):
    s = json.load(open(f"{out}/{plan}/pmc_summary.json"))["summary"]
    for k, v in s.items():
        if plan == "semi" and k not in ("k1_paths", "kf_lean"): continue
        if plan == "unfused" and k not in ("k2_eval_book", "k4_cva_paths"): continue
        res[f"{plan}:{k}"] = {"kernel": v["kernel"][:90], "hbm_bytes_measured": v["hbm_bytes_per_launch_at_1Mi_paths"], "algorithmic_bytes": alg[k],
                              "ratio": v["hbm_bytes_per_launch_at_1Mi_paths"] / alg[k]}
json.dump(res, open(f"{out}/plans_traffic.json", "w"), indent=1)
for k, v in res.items(): print(k, "measured %.3f GB  algorithmic %.3f GB  ratio %.3f" % (v["hbm_bytes_measured"] / 1e9, v["algorithmic_bytes"] / 1e9, v["ratio"]))
PY
