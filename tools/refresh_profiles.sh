#!/bin/bash
# usage (on the GPU box): tools/refresh_profiles.sh   -> everything under gpurun_out/rp_*
# the round's measurement set: rocprofv3 summaries (ViT-B overlapped + serial, CaiT, Swin), PMC
# traffic / MFMA-busy of the headline run, and one bench line per configuration
set -e
tools/prof.sh rp_vitb --steps 5 --warmup 2
VITMI_OVERLAP_WGRAD=0 tools/prof.sh rp_vitb_serial --steps 5 --warmup 2 --graph off
tools/pmc.sh rp_vitb --steps 3 --warmup 1
for cfg in "cait_S24_224 64" "cait_S24_224 256" "swin_tiny_patch4_window7_224 128" "swin_tiny_patch4_window7_224 256"; do
  set -- $cfg
  tools/prof.sh rp_$1_bs$2 --arch $1 --batch $2 --steps 3 --warmup 1 --graph off
  python bench.py --arch $1 --batch $2 > gpurun_out/rp_bench_$1_bs$2.json 2>/dev/null
done
python bench.py > gpurun_out/rp_bench_vitb.json 2>gpurun_out/rp_bench_vitb.err
python bench.py --arch dino_vitb8 --img 96 --batch 128 > gpurun_out/rp_bench_vitb8.json 2>/dev/null
python bench.py --arch dino_vits16 --img 32 --batch 128 > gpurun_out/rp_bench_vits16.json 2>/dev/null
for a in dino_vitb16 swin_tiny_patch4_window7_224 cait_S24_224; do
  python bench.py --arch $a --mode lineareval > gpurun_out/rp_bench_lineareval_$a.json 2>/dev/null
done
python tools/bench_print.py gpurun_out/rp_bench_*.json
