#!/bin/bash
O=$PWD/gpurun_out/r3y; mkdir -p $O
for b in ee pv; do
  timeout -k 10 300 python tools/large_book.py --book $b --repeat 3 --profile > $O/large_book_$b.json 2> $O/large_book_$b.err || { tail -5 $O/large_book_$b.err; exit 1; }
  python3 -c "
import json
for l in open('$O/large_book_$b.json'):
    d=json.loads(l); print('$b run_s %.3f  products/s %.0f  %s retries %s'%(d['run_s'], d['products_per_second'], d['prepare'], d['lsm_singular_retries']))"
  grep -A 22 'Ordered by' $O/large_book_$b.err | cut -c1-150
done
