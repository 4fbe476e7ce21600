#!/bin/bash
O=$PWD/gpurun_out/r3q; mkdir -p $O
timeout -k 10 900 bash tools/measure_counters.sh $O/counters_fused --no-strong --sustain 0 --plan fused > $O/counters_fused.log 2>&1; tail -3 $O/counters_fused.log | cut -c1-200
