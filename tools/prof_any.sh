#!/bin/bash
# usage (on the GPU box): tools/prof_any.sh <tag> <python script> [args...]: rocprofv3 kernel stats of any script -> gpurun_out/<tag>_kernel_stats.csv
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
script=$1; shift
case $script in /*) ;; *) script=$R/$script ;; esac
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 "$script" "$@" > $R/gpurun_out/$tag.log 2>&1
cd $R
cp $(find gpurun_out/$tag -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/$tag
python3 - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")))[:16]:
    print(f'{r["Name"][:72]:72s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]} %')
PY
