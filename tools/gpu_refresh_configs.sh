#!/bin/bash
# refresh of the configuration-level artefacts only (the bench / counter set comes from gpu_final.sh)
O=$PWD/gpurun_out/${1:-refresh}; mkdir -p $O
timeout -k 10 600 python tools/run_configs.py 2 3 4 5 6 > $O/configs.jsonl 2> $O/configs.err; echo "configs rc=$?"
timeout -k 10 600 python tools/large_book.py --repeat 3 > $O/large_book.json 2> $O/large_book.err; echo "large book rc=$?"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 $GRAFT_REPO_ROOT/tools/prof_cfg5.py > $O/prof_cfg5.json 2> /dev/null ); echo "rocprof cfg5 rc=$?"
cp $(find $O/prof_cfg5 -name "*kernel_stats.csv" | head -1) $O/config5_kernel_stats.csv
