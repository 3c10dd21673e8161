#!/bin/bash
# usage (GPU box): tools/lib_ab_models.sh <other.so> [rounds]  -> bench.py of Swin-T and CaiT-S24 (bs 256) with <other.so> and with the tree's library, interleaved
other=$1; n=${2:-2}
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d["value"], d["ms_per_step"])'
for a in swin_tiny_patch4_window7_224 cait_S24_224; do
  for r in $(seq $n); do
    echo -n "$a other: "; VITMI_LIB=$other python bench.py --arch $a --batch 256 --lean --no-cpu-baseline 2>/dev/null | python3 -c "$P"
    echo -n "$a tree:  "; python bench.py --arch $a --batch 256 --lean --no-cpu-baseline 2>/dev/null | python3 -c "$P"
  done
done
