#!/bin/bash
O=$PWD/gpurun_out/r4c; mkdir -p $O
timeout -k 10 300 python tools/check_heston_greeks.py > $O/heston_greeks.json 2> $O/heston_greeks.err || { tail -5 $O/heston_greeks.err; exit 1; }
cat $O/heston_greeks.json
