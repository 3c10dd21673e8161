#!/bin/bash
# usage (on the GPU box): tools/pmc_kernel.sh <tag> <python script> [args...] -> gpurun_out/<tag>_pmc_kernel.txt
# SQ counters per kernel of an arbitrary tool script (one --pmc pass per group, no trace domains)
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_k$i -- python3 $R/"$@" > $R/gpurun_out/${tag}_k$i.log 2>&1 || echo "pass $i ($grp) failed"
  cd $R
done
python3 tools/pmc_counters.py gpurun_out/${tag}_pmc_kernel.txt gpurun_out/${tag}_k*/ || true
rm -rf gpurun_out/${tag}_k*/
