#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <tag> [bench args...]  -> gpurun_out/<tag>_pmc_traffic.json
# two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same bench command, no trace domains
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_fetch -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_write -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_write.log 2>&1
cd $R
python3 tools/pmc_traffic.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_pmc_traffic.txt
rm -rf gpurun_out/${tag}_fetch gpurun_out/${tag}_write
# third pass: matrix-pipe busy cycles against the chip's active cycles (MFMA-busy fraction)
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${tag}_mfma -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_mfma.log 2>&1 || true
cd $R
python3 tools/pmc_mfma.py gpurun_out/${tag}_mfma gpurun_out/${tag}_pmc_mfma.json ${PMC_CLOCK_MHZ:-} > gpurun_out/${tag}_pmc_mfma.txt || true
rm -rf gpurun_out/${tag}_mfma
