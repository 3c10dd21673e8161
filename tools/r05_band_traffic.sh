#!/bin/bash
# usage (on the GPU box): tools/r05_band_traffic.sh   -> gpurun_out/r05_band{default,off}_gemm_traffic_by_shape.txt
# the per-shape PMC traffic table of tools/pmc_gemm_shapes.sh twice: with the column-band tile order (default) and with
# every shape in row-major order (VITMI_GEMM_BAND_KB=1000000): the counted over-fetch each order costs.
set -e
rm -f gpurun_out/r05_bandoff_shapes.txt gpurun_out/r05_banddefault_shapes.txt
VITMI_GEMM_BAND_KB=1000000 tools/pmc_gemm_shapes.sh r05_bandoff
tools/pmc_gemm_shapes.sh r05_banddefault
tail -3 gpurun_out/r05_bandoff_gemm_traffic_by_shape.txt gpurun_out/r05_banddefault_gemm_traffic_by_shape.txt
