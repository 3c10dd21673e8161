#!/bin/bash
# usage: tools/sweep_bench.sh <outfile> "<hooks A>" "<hooks B>" ...   each = space-separated hook=value list for tools/bench_with.py
# one bench.py run per setting (fresh process each), interleaved twice; prints ms_per_step / gemm ms
out=$1; shift
: > $out
for rnd in 1 2; do
  for h in "$@"; do
    python tools/bench_with.py $h -- --lean --no-cpu-baseline --no-configs --no-dp-proxy --no-fp32-rate 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$h', 'round $rnd', d['ms_per_step'], d['roofline'].get('gemm_ms_per_step'))
" >> $out
  done
done
cat $out
