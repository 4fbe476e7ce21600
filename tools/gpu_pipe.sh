#!/bin/bash
# pipelined passes: parity test, the bench line without and with a one-rank RCCL group
O=gpurun_out/${1:-pipe}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "rccl or comm" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -B30 "short test summary" $O/pytest.log | head -60
show() { python -c "import json,sys;d=json.loads(open(sys.argv[1]).read().strip().split(chr(10))[-1]);print(sys.argv[1],d['ms_per_step'],d['roofline']['kernel_ms'],d['config']['passes_in_flight'],d['result']['cva'])" $1; }
for i in 1 2; do
MCX_BENCH_NO_PIPELINE=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_plain_nopipe_$i.json 2> $O/err; show $O/bench_plain_nopipe_$i.json
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_plain_pipe_$i.json 2> $O/err; show $O/bench_plain_pipe_$i.json
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline > $O/bench_tr1.json 2> $O/bench_tr1.err; echo "torchrun1 rc=$?"; show $O/bench_tr1.json
MCX_BENCH_NO_PIPELINE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --no-cpu-baseline > $O/bench_tr1_nopipe.json 2> $O/bench_tr1.err; echo "torchrun1 nopipe rc=$?"; show $O/bench_tr1_nopipe.json
