#!/bin/bash
# usage (on the GPU box): tools/refresh_profiles.sh [tag]   -> everything under gpurun_out/<tag>_*
# the round's measurement set: rocprofv3 summaries (ViT-B as benchmarked + graph off, CaiT, Swin), PMC
# traffic / MFMA-busy of the headline run, per-shape GEMM traffic, one bench line per configuration.
# tools/collect_profiles.sh <tag> copies the judged files into profiles/.
set -e
T=${1:-r03}
Q="--no-parity --no-alt"
python bench.py > gpurun_out/${T}_bench_n1.json 2>gpurun_out/${T}_bench_n1.err
tools/prof.sh ${T}_vitb --steps 5 --warmup 2 $Q
tools/prof.sh ${T}_vitb_serial --steps 5 --warmup 2 --graph off $Q
tools/pmc.sh ${T}_vitb --steps 3 --warmup 1 $Q
python bench.py --residual fp32 --no-cpu-baseline > gpurun_out/${T}_bench_n1_residual_fp32.json 2>/dev/null
python bench.py --arch dino_vitb8 --img 96 --batch 128 > gpurun_out/${T}_bench_n1_dino_vitb8_96_bs128.json 2>/dev/null
python bench.py --arch dino_vits16 --img 32 --batch 128 > gpurun_out/${T}_bench_n1_dino_vits16_32_bs128.json 2>/dev/null
for cfg in "cait_S24_224 256" "swin_tiny_patch4_window7_224 256"; do
  set -- $cfg
  tools/prof.sh ${T}_$1_bs$2 --arch $1 --batch $2 --steps 3 --warmup 1 --graph off $Q
  python bench.py --arch $1 --batch $2 > gpurun_out/${T}_bench_n1_$1_bs$2.json 2>/dev/null
done
for a in dino_vitb16 swin_tiny_patch4_window7_224 cait_S24_224; do
  python bench.py --arch $a --mode lineareval > gpurun_out/${T}_bench_n1_lineareval_$a.json 2>/dev/null
done
rm -f gpurun_out/${T}_shapes.txt
tools/pmc_gemm_shapes.sh ${T}
tools/clock_check.sh ${T}_vitb 32
python tools/bench_print.py gpurun_out/${T}_bench_n1*.json
