#!/bin/bash
# does the power-of-two leading dimension of the paths tensor cost the streaming kernels bandwidth?  semi + unfused plans at 2^20 paths
# and at path counts whose row stride is not a power of two
O=$PWD/gpurun_out/r3w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for p in 1048576 1049088 1050624 1056768 1000000; do for plan in semi unfused; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${plan}_$p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-strong --sustain 0 --plan $plan --paths $p --steps 60 > $O/${plan}_$p.json 2> $O/${plan}_$p.err || { echo "$p failed"; tail -3 $O/${plan}_$p.err; exit 1; }
  f=$(find $O/prof_${plan}_$p -name "*kernel_stats.csv" | head -1)
  echo "== $plan $p"; grep -v "rocclr\|kf_merge\|k4_finish\|Name" $f | awk -F'",' '{split($2,a,","); printf("   %-70s avg %.1f us  per-path %.4f ns\n", substr($1,2,70), a[3]/1000, a[3]/'$p')}'
done; done
