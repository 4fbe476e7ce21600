#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2f
timeout -k 10 120 ./tools/ubench_valu > gpurun_out/r2f/ubench_valu.txt 2>&1; echo "ubench rc=$?"; cat gpurun_out/r2f/ubench_valu.txt
timeout -k 10 600 tools/measure_counters.sh $PWD/gpurun_out/r2f/counters --plan fused > gpurun_out/r2f/counters.log 2>&1; echo "counters rc=$?"; tail -3 gpurun_out/r2f/counters.log
cp gpurun_out/r2f/counters/counters.json profiles/counters.json 2>/dev/null
timeout -k 10 600 python bench.py > gpurun_out/r2f/bench_default.json 2> gpurun_out/r2f/bench_default.err; echo "bench rc=$?"; cat gpurun_out/r2f/bench_default.json; tail -3 gpurun_out/r2f/bench_default.err
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2f/bench_torchrun1.json 2> gpurun_out/r2f/bench_torchrun1.err; echo "torchrun1 rc=$?"; cat gpurun_out/r2f/bench_torchrun1.json; tail -5 gpurun_out/r2f/bench_torchrun1.err
timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2f/bench_gpus2.json 2> gpurun_out/r2f/bench_gpus2.err; echo "gpus2 (expected to fail on a 1-GPU box) rc=$?"; tail -5 gpurun_out/r2f/bench_gpus2.err
