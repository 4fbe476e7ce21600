#!/bin/bash
# round 3: hardened VALU microbenchmark + shader clock during it, counters of the bench kernel (SQ mix, HBM traffic) -> counters.json
O=$PWD/gpurun_out/r3q; mkdir -p $O
timeout -k 10 300 ./tools/ubench_valu > $O/ubench_valu.txt 2>&1; cat $O/ubench_valu.txt
timeout -k 10 400 bash tools/measure_clock.sh $O/clock > $O/clock.log 2>&1; tail -16 $O/clock.log
timeout -k 10 900 bash tools/measure_counters.sh $O/counters --no-strong --sustain 0 > $O/counters.log 2>&1; tail -5 $O/counters.log | cut -c1-300
