"""Which kernels surround each device-to-device copy of a replayed step: tools/copy_neighbours.py <kernel_trace.csv>"""
import csv, re, sys
from collections import Counter
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", "")) for r in rows))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n); return n[:48]
cuts = [i for i, k in enumerate(ks) if "sgd_momentum" in k[2]]
steps = [ks[a + 1:b + 1] for a, b in zip(cuts[:-1], cuts[1:])][-4:-1]
c = Counter()
for s in steps:
    for i, k in enumerate(s):
        if "copyBuffer" in k[2] or "elementwise" in k[2] or "Fill" in k[2]:
            prev = short(s[i - 1][2]) if i else "-"; nxt = short(s[i + 1][2]) if i + 1 < len(s) else "-"
            c[(short(k[2]), k[3], (k[1] - k[0]) // 100 / 10, prev, nxt)] += 1
for k, v in sorted(c.items(), key=lambda kv: -kv[1]): print(v / len(steps), k)
