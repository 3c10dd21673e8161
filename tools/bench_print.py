"""Print value / ms_per_step / graph flag / GEMM TFLOP/s of bench.py JSON lines."""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(f, d["value"], d["ms_per_step"], d["config"].get("hip_graph"), r.get("achieved"))
