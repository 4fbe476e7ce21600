#!/bin/bash
# round 3, final measurement set, part 1: counters of the bench kernel at HEAD (-> profiles/counters.json on the box, so that the bench
# lines below carry `traffic` and `alu`), the GPU suite, the bench at the strong-scaling path counts, the default line, the one-rank
# RCCL line, kernel stats of the default run
O=$PWD/gpurun_out/r3final; mkdir -p $O
timeout -k 10 700 bash tools/measure_counters.sh $O/counters --no-strong --sustain 0 --plan fused > $O/counters.log 2>&1 || { tail -5 $O/counters.log; exit 1; }
cp $O/counters/counters.json profiles/counters.json; echo "counters: sha $(python3 -c "import json;print(json.load(open('profiles/counters.json'))['kernel_source_sha'])")"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
for p in 131072 262144 524288 1048576; do
  timeout -k 10 120 python bench.py --paths $p --no-cpu-baseline --no-strong --sustain 0 --plan fused --steps 40 > $O/bench_paths_$p.json 2> $O/bench_paths_$p.err || exit 1
  python3 - $O/bench_paths_$p.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("paths %d  ms/step %.4f kernel_ms %.4f value %.4e cva %.10f" % (d["config"]["paths_per_gpu"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["value"], d["result"]["cva"]), flush=True)
PY
done
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 - $O/bench_default.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("default: ms/step %.4f kernel_ms %.4f value %.4e frac %.4f alu %s strong %s sustained %s cpu %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["value"], d["roofline"]["frac"], (d.get("alu") or {}).get("frac"), d["strong"], d["sustained"], d.get("cpu_baseline")))
PY
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline > $O/bench_rccl1.json 2> $O/bench_rccl1.err; python3 - $O/bench_rccl1.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("one-rank RCCL: ms/step %.4f kernel_ms %.4f value %.4e strong %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["value"], d["strong"]))
PY
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-strong --steps 200 > $O/bench_under_rocprof.json 2> $O/prof.err ); echo "prof rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/bench_kernel_stats.csv; head -4 $O/bench_kernel_stats.csv | cut -c1-200
