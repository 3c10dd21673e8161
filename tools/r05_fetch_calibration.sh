#!/bin/bash
# usage (on the GPU box): tools/r05_fetch_calibration.sh  -> gpurun_out/r05_fetch_calibration.txt
# Calibrates rocprofv3's FETCH_SIZE on the GEMM kernels' OWN access patterns (the guide: "Other access widths are
# uncalibrated: calibrate on a known byte count in your own access pattern").  One column tile (N = 256): every A byte is
# needed by exactly one workgroup, so the true read traffic of a call is A once + the 256-column B panel per XCD.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/r05_fetch_calibration.txt
for s in nt:50432:256:768 nn:50432:256:768 nt:50432:256:3072 nn:50432:256:3072 tn:256:256:50432:store:f32; do
  rm -rf $R/gpurun_out/r05_cal
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r05_cal -- python3 $R/tools/gemm_bench.py $s > $R/gpurun_out/r05_cal.log 2>&1
  python3 - "$s" "$R/gpurun_out/r05_cal" >> $R/gpurun_out/r05_fetch_calibration.txt <<'PY'
import csv, glob, sys
spec, d = sys.argv[1], sys.argv[2]
p = spec.split(":"); lay, M, N, K = p[0], int(p[1]), int(p[2]), int(p[3])
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
tot = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and ("gemm_fast" in r["Kernel_Name"] or "splitk_reduce" in r["Kernel_Name"]):
        k = "gemm" if "gemm_fast" in r["Kernel_Name"] else "reduce"
        tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
calls = 23
a_bytes = M * K * 2 if lay != "tn" else K * M * 2
b_once = N * K * 2
raw = tot.get("gemm", 0.0) * 1024 / calls
print(f"{spec:28s} raw FETCH_SIZE {raw/1e6:8.1f} MB/call   A {a_bytes/1e6:7.1f} MB + B {b_once/1e6:5.2f} MB x 8 XCDs = {(a_bytes + 8*b_once)/1e6:7.1f} MB   raw / true = {raw/(a_bytes + 8*b_once):.3f}")
PY
done
rm -rf $R/gpurun_out/r05_cal
cat $R/gpurun_out/r05_fetch_calibration.txt
