set -e
A="--arch swin_tiny_patch4_window7_224 --batch 256 --lean --no-cpu-baseline --graph off"
for r in 1 2; do
for v in 0 1; do
python tools/bench_with.py win_bwd_prefetch=$v -- $A 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('prefetch $v round $r', d['ms_per_step'], d['value'])
"
done
done
