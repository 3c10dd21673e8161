"""Whole-step HBM rate from a tools/pmc.sh traffic file and the step time of a bench line:
sum over kernels of (2 x FETCH_SIZE + WRITE_SIZE) per step / ms per step  (C4 / C5 are judged on HBM GB/s).
usage: hbm_rate.py <pmc_traffic.json> <bench.json> [peak TB/s = 8.0]"""
import json, sys
pm = json.load(open(sys.argv[1]))
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
peak = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
steps = (pm.get("_meta") or {}).get("steps_profiled")
tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in pm.items() if k != "_meta")
per_step = tot / steps
ms = d["ms_per_step"]
print(f"{d['config']['workload'].split(', 10 classes')[0]}: {per_step / 1e9:.2f} GB of HBM traffic per step (PMC, {steps} steps profiled), "
      f"{ms:.2f} ms/step -> {per_step / ms / 1e9:.2f} TB/s = {per_step / ms / 1e9 / peak:.3f} of the {peak:.0f} TB/s peak")
top = sorted(((v["hbm_bytes_per_launch"] * v["launches"] / steps, k) for k, v in pm.items() if k != "_meta"), reverse=True)[:8]
for b, k in top:
    print(f"   {b / 1e9:7.2f} GB/step  {k[:70]}")
