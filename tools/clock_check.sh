#!/bin/bash
# usage (on the GPU box): tools/clock_check.sh <tag>   -> gpurun_out/<tag>_clock.txt
# tools/clock_probe.py under rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES (no other trace domain):
# method 2 (GRBM_GUI_ACTIVE / 8 / wall) next to method 1 (in-kernel stamps) on the same >= 10 ms dispatches.
set -e
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/clk_$$
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/clk_$$ -- python3 $R/tools/clock_probe.py ${2:-32} > $R/gpurun_out/${tag}_clock_probe.log 2>&1
cd $R
python3 - /tmp/clk_$$ gpurun_out/${tag}_clock_probe.log > gpurun_out/${tag}_clock.txt <<'EOF'
import csv, glob, json, sys
d, log = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
cnt = {}
rows = list(csv.DictReader(open(cc)))
print("counter csv columns:", list(rows[0].keys()))
for r in rows:                                   # several rows per (dispatch, counter) are summed (per-XCD / per-SE dimensions)
    c = cnt.setdefault(r["Dispatch_Id"], {})
    c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
probes = [json.loads(l) for l in open(log) if l.startswith("{")]
print("method 1 (in-kernel s_memtime / s_memrealtime, median over workgroups) vs method 2 (GRBM_GUI_ACTIVE / 8 / wall), per shape;")
print("MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x wall x clock), dispatches of >= 10 ms only")
big = [(k, v) for k, v in dur.items() if "gemm_fast_kernel" in v[1] and k in cnt]
big.sort(key=lambda kv: int(kv[0]))
big = big[-7 * len(probes):]                       # the probe's own dispatches (7 per shape) follow the warm-up launches
# per shape: 3 warm + 3 timed + 1 stamped dispatches, in order
for i, p in enumerate(probes):
    grp = big[i * 7:(i + 1) * 7][3:6]
    if not grp:
        continue
    clk2 = [cnt[k]["GRBM_GUI_ACTIVE"] / 8.0 / v[0] * 1e3 for k, v in grp]          # cycles per ns -> MHz
    busy = [cnt[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * v[0] * 1e-9 * p["clock_mhz_in_kernel"] * 1e6) for k, v in grp]
    c2 = sorted(clk2)[1]
    print(f"{p['shape']:28s} M {p['M']}: wall {grp[1][1][0] / 1e6:7.2f} ms  {p['tflops']:7.1f} TFLOP/s  in-kernel {p['clock_mhz_in_kernel']:7.1f} MHz  "
          f"GRBM {c2:7.1f} MHz  ratio {c2 / p['clock_mhz_in_kernel']:.3f}  MFMA-busy {sorted(busy)[1] * 100:5.1f} %")
EOF
rm -rf /tmp/clk_$$
cat gpurun_out/${tag}_clock.txt
