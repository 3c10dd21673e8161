"""Summarise a rocprofv3 --kernel-trace --stats run: per-kernel totals per step."""
import csv, glob, re, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = d if d.endswith(".csv") else glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# the steps actually executed (graph trial steps, capture warm-ups and the instrumented step
# included) = launches of a once-per-step kernel, when the trace holds one
for r in rows:
    if re.search(r"xent_kernel\(|xent_kernel[A-Z]", r["Name"]) and "reduce" not in r["Name"]:
        steps = float(r["Calls"])
        break
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over {steps:g} steps = {tot/1e6/steps:.2f} ms/step")
print(f"{'ms/step':>9} {'calls/step':>10} {'avg us':>9} {'%':>6}  kernel")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)[:100]
    print(f"{float(r['TotalDurationNs'])/1e6/steps:9.3f} {int(r['Calls'])/steps:10.1f} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}  {n}")

# ---- the four buckets VERDICT r04 asked for (per step), and a JSON of them for bench.py's `roofline.frac_rocprof`
BUCKETS = [("gemm_main", r"gemm_fast_kernel|gemm_fast2_kernel|gemm_generic_kernel|gemm_skinny_kernel|gemm_small_kernel"),
           ("gemm_finishers", r"splitk_reduce|tail_epilogue"),
           ("attention", r"attn_|th_pack|class_attn|win_attn|relpos"),
           ("layernorm", r"ln_fwd|ln_bwd"),
           ("optimizer", r"sgd_momentum|adam_kernel|adagrad|adadelta|adabelief")]
acc = {k: 0.0 for k, _ in BUCKETS}
acc["other"] = 0.0
for r in rows:
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    for k, pat in BUCKETS:
        if re.search(pat, r["Name"]):
            acc[k] += ms
            break
    else:
        acc["other"] += ms
print("buckets (ms/step): " + "  ".join(f"{k} {v:.3f}" for k, v in acc.items()))
if len(sys.argv) > 4:
    import json, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        import bench
        sha = bench.kernel_sources_sha256()
    except Exception:
        sha = None
    json.dump({"_meta": {"kernel_sources_sha256": sha, "steps_profiled": steps, "source": os.path.basename(f)},
               "ms_per_step_kernel_time": tot / 1e6 / steps, "buckets_ms_per_step": acc}, open(sys.argv[4], "w"), indent=1)
