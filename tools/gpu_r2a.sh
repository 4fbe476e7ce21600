#!/bin/bash
# round-2 measurement call A: issue-cost micro-benchmark, parity of the new straight-line kernel, A/B against the round-1 library
set -o pipefail
mkdir -p gpurun_out/r2a
timeout -k 10 120 ./tools/ubench_valu > gpurun_out/r2a/ubench_valu.txt 2>&1; echo "ubench rc=$?"
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q > gpurun_out/r2a/pytest_parity.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r2a/pytest_parity.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --plan fused --steps 30 > gpurun_out/r2a/bench_new_$i.json 2> gpurun_out/r2a/bench_new_$i.err || exit 1
MCX_LIB_PATH=$PWD/variants/libmcx_r01.so timeout -k 10 300 python bench.py --no-cpu-baseline --plan fused --steps 30 > gpurun_out/r2a/bench_old_$i.json 2> gpurun_out/r2a/bench_old_$i.err || exit 1
done
MCX_LIB_PATH=$PWD/variants/libmcx_r01.so timeout -k 10 300 python bench.py --no-cpu-baseline --plan fused --steps 30 --paths 917504 > gpurun_out/r2a/bench_old_tail.json 2> gpurun_out/r2a/bench_old_tail.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2a/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, 'ms/step %.4f'%d['ms_per_step'], 'value %.4g'%d['value'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'cva', d['result'])
    except Exception as e: print(f, 'ERR', e)
PY
