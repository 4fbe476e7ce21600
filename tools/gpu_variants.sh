#!/bin/bash
O=$PWD/gpurun_out/variants; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_book_kernel_variants.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 -m pytest $GRAFT_REPO_ROOT/tests/test_book_kernel_variants.py -m gpu -x -q > $O/pytest_prof.log 2>&1 ); echo "prof rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/book_variants_kernel_stats.csv; grep -i "k2_eval\|k4_cva" $f | cut -c1-160
