#!/bin/bash
# usage (on the GPU box): tools/pmc_gemm_shapes.sh <tag>
# Per-shape HBM traffic of the twelve GEMMs of a ViT-B/16 block (bs 256), each with its real epilogue
# (bf16 residual stream, the benchmarked mode):
# two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) per shape, no trace domains; summarised
# against the algorithmic bytes by tools/pmc_gemm_shapes.py -> gpurun_out/<tag>_gemm_traffic_by_shape.{txt,json}
set -e
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
SHAPES="nt:50432:2304:768 nt:50432:3072:768:gelu nt:50432:768:768:res:bf16 nt:50432:768:3072:res:bf16 \
nn:50432:3072:768:dgelu nn:50432:768:3072 nn:50432:768:2304 nn:50432:768:768 \
tn:768:768:50432:store:f32 tn:3072:768:50432:store:f32 tn:768:3072:50432:store:f32 tn:2304:768:50432:store:f32"
cd /tmp && export TMPDIR=/tmp
i=0
for s in $SHAPES; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${tag}_shape${i}_$c -- python3 $R/tools/gemm_bench.py $s > $R/gpurun_out/${tag}_shape${i}_$c.log 2>&1
  done
  echo "$i $s" >> $R/gpurun_out/${tag}_shapes.txt
  i=$((i+1))
done
cd $R
python3 tools/pmc_gemm_shapes.py gpurun_out/${tag} > gpurun_out/${tag}_gemm_traffic_by_shape.txt
rm -rf gpurun_out/${tag}_shape*_FETCH_SIZE gpurun_out/${tag}_shape*_WRITE_SIZE
