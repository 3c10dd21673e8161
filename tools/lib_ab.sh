#!/bin/bash
# usage (GPU box): tools/lib_ab.sh <other.so> [rounds]   -> tools/step_shapes_bench.py with the tree's library and with <other.so>, interleaved
other=$1; n=${2:-2}
for r in $(seq $n); do
  VITMI_LIB=$other python tools/step_shapes_bench.py 2>/dev/null
  python tools/step_shapes_bench.py 2>/dev/null
done
