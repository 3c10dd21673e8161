"""Per-kernel averages of arbitrary rocprofv3 --pmc counters (one or more passes' output directories).
usage: pmc_counters.py <out.txt> <dir> [<dir> ...]     prints counter / launch for the 12 heaviest kernels"""
import csv, glob, re, sys
from collections import defaultdict

val = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", k)
            k = re.split(r"[<(]|I[Lb]|ID|If", k)[0].replace("void ", "")
            val[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
names = sorted({c for k in val for c in val[k]})
order = sorted(val, key=lambda k: -val[k].get("SQ_WAVE_CYCLES", val[k].get(names[0], 0)))[:14]
with open(sys.argv[1], "w") as out:
    for k in order:
        line = f"{k[:44]:44s} " + "  ".join(f"{c}={val[k][c] / max(cnt[k][c], 1):.4g}" for c in names if c in val[k])
        print(line); out.write(line + "\n")
