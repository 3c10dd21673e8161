"""Idle time between consecutive kernels of a step: tools/gap_report.py <kernel_trace.csv> [top].
Reads a rocprofv3 --kernel-trace CSV (Start_Timestamp / End_Timestamp in ns), cuts the trace into steps at
xent_kernel, and reports for the LAST steps (the timed, graph-replayed ones): wall time first start -> last end,
sum of kernel durations, sum and count of gaps, and which kernel pairs the largest total gap sits between."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 15
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:70]


cuts = [i for i, k in enumerate(ks) if "sgd_momentum" in k[2]]
steps = [ks[a + 1:b + 1] for a, b in zip(cuts[:-1], cuts[1:])]
steps = steps[-5:]
for s in steps:
    wall = s[-1][1] - s[0][0]
    busy = sum(e - b for b, e, _ in s)
    gaps = [max(0, s[i + 1][0] - s[i][1]) for i in range(len(s) - 1)]
    print(f"step: {len(s)} kernels, wall {wall/1e6:.3f} ms, kernel time {busy/1e6:.3f} ms, gaps {sum(gaps)/1e6:.3f} ms "
          f"(median {sorted(gaps)[len(gaps)//2]/1e3:.2f} us, max {max(gaps)/1e3:.1f} us)")
acc = defaultdict(lambda: [0, 0])
for s in steps:
    for i in range(len(s) - 1):
        g = max(0, s[i + 1][0] - s[i][1])
        k = (short(s[i][2]), short(s[i + 1][2]))
        acc[k][0] += g
        acc[k][1] += 1
print("largest gap totals per step (us, count, avg us): after -> before")
for k, (g, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{g/1e3/len(steps):8.1f} {c/len(steps):6.1f} {g/1e3/c:6.2f}  {k[0]}  ->  {k[1]}")
