#!/bin/bash
# round-end measurement call: counters of the bench kernels, rocprofv3 kernel stats of the bench command, the bench line, the
# BASELINE configurations at full size, the large book.  Everything lands under gpurun_out/<tag>/ (copy the summaries to profiles/).
TAG=${1:-final}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 tools/measure_counters.sh $O/counters --plan fused > $O/counters.log 2>&1; echo "counters rc=$?"
cp $O/counters/counters.json profiles/counters.json
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err ); echo "rocprof bench rc=$?"
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
timeout -k 10 600 python tools/run_configs.py 2 3 4 5 6 > $O/configs.jsonl 2> $O/configs.err; echo "configs rc=$?"
timeout -k 10 600 python tools/large_book.py > $O/large_book.json 2> $O/large_book.err; echo "large book rc=$?"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_unfused -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --plan unfused --steps 10 > $O/bench_unfused.json 2> /dev/null ); echo "rocprof unfused rc=$?"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_semi -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --plan semi --steps 10 > $O/bench_semi.json 2> /dev/null ); echo "rocprof semi rc=$?"
