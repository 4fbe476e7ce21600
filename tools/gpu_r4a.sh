#!/bin/bash
# bump-and-revalue of config 3: with the host-thread guard (default) and without it (unguarded); process CPU time and CFS throttling
O=$PWD/gpurun_out/r4a; mkdir -p $O
cat /sys/fs/cgroup/cpu.max 2>/dev/null
for m in default unguarded; do
  timeout -k 10 300 python tools/prof_bump.py $m > $O/prof_bump_$m.txt 2> $O/prof_bump_$m.err || { tail -5 $O/prof_bump_$m.err; exit 1; }
  echo "== $m"; grep "threads\|bumps\|process CPU" $O/prof_bump_$m.txt | cut -c1-220
done
timeout -k 10 300 python tools/run_configs.py 6 > $O/config6.jsonl 2> $O/config6.err; cut -c1-400 $O/config6.jsonl
