#!/bin/bash
# usage (on the GPU box): tools/prof.sh <tag> [bench args...]
# rocprofv3 kernel trace + stats of bench.py; summary -> gpurun_out/<tag>.txt, CSV kept as
# gpurun_out/<tag>_kernel_stats.csv, bench JSON line -> gpurun_out/<tag>.json
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/$tag.log 2>&1
cd $R
grep '^{"metric"' gpurun_out/$tag.log > gpurun_out/$tag.json || true
steps=$(python3 -c "import json;d=json.load(open('gpurun_out/$tag.json'));print(d['steps']+d['warmup']+1)")
python3 tools/prof_summary.py gpurun_out/$tag $steps 30 gpurun_out/${tag}_buckets.json > gpurun_out/$tag.txt
cp $(find gpurun_out/$tag -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/$tag
