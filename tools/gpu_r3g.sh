#!/bin/bash
# SQ counters of the one-launch kernel at a small path count (shape via variants/libmcx_ab.so) and at full size
export MCX_LIB_PATH=$PWD/variants/libmcx_ab.so
O=$PWD/gpurun_out/r3g
MCX_LEAN_SHAPE=11 bash tools/measure_sq.sh $O/s11_131072 --paths 131072 --plan fused --no-strong --sustain 0 --clock-warmup 0.2 > $O.s11.log 2>&1
MCX_LEAN_SHAPE=21 bash tools/measure_sq.sh $O/s21_131072 --paths 131072 --plan fused --no-strong --sustain 0 --clock-warmup 0.2 > $O.s21.log 2>&1
MCX_LEAN_SHAPE=21 bash tools/measure_sq.sh $O/s21_1048576 --paths 1048576 --plan fused --no-strong --sustain 0 --clock-warmup 0.2 > $O.s21f.log 2>&1
tail -4 $O.s11.log; tail -4 $O.s21.log; tail -4 $O.s21f.log
