#!/bin/bash
# large book after the host-side work (one-call batched LSM, vectorised event layout, in-place DevEvents, pinned coefficient readback)
O=$PWD/gpurun_out/r3t; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/large_book.py --repeat 4 --profile > $O/large_book.json 2> $O/large_book.err || { tail -5 $O/large_book.err; exit 1; }
cut -c1-1000 $O/large_book.json
grep -A 28 'Ordered by' $O/large_book.err | cut -c1-150
