#!/bin/bash
# usage (on the GPU box): tools/pmc_extra.sh <tag> [bench args...] -> gpurun_out/<tag>_pmc_extra.txt
# a few SQ / TCC counters per kernel, one --pmc pass per group (no trace domains)
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_x$i -- python3 $R/bench.py --no-cpu-baseline --graph off "$@" > $R/gpurun_out/${tag}_x$i.log 2>&1 || echo "pass $i ($grp) failed"
  cd $R
done
python3 tools/pmc_counters.py gpurun_out/${tag}_pmc_extra.txt gpurun_out/${tag}_x*/ || true
rm -rf gpurun_out/${tag}_x*/
