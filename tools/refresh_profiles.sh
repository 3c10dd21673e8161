#!/bin/bash
# usage (on the GPU box): tools/refresh_profiles.sh [tag]   -> everything under gpurun_out/<tag>_*
# the round's measurement set: rocprofv3 summaries (ViT-B as benchmarked + graph off, CaiT, Swin), PMC
# traffic / MFMA-busy of the headline run, per-shape GEMM traffic, one bench line per configuration.
# tools/collect_profiles.sh <tag> copies the judged files into profiles/.
set -e
T=${1:-r04}
PART=${2:-all}          # a = headline set, b = other configurations, c = lineareval / per-shape traffic / clock / switches (a call is capped at 20 min)
Q="--lean"
if [ $PART = all ] || [ $PART = a ]; then
python bench.py > gpurun_out/${T}_bench_n1.json 2>gpurun_out/${T}_bench_n1.err
tools/prof.sh ${T}_vitb --steps 5 --warmup 2 $Q
tools/prof.sh ${T}_vitb_serial --steps 5 --warmup 2 --graph off $Q
# MFMA-busy at the in-kernel clock the bench line of this same box reports
export PMC_CLOCK_MHZ=$(python3 -c "import json;print(json.loads(open('gpurun_out/${T}_bench_n1.json').read().strip().splitlines()[-1])['roofline'].get('clock_mhz_under_load') or '')")
tools/pmc.sh ${T}_vitb --steps 3 --warmup 1 $Q
unset PMC_CLOCK_MHZ
python bench.py --residual fp32 --no-cpu-baseline $Q > gpurun_out/${T}_bench_n1_residual_fp32.json 2>/dev/null
python tools/hbm_rate.py gpurun_out/${T}_vitb_pmc_traffic.json gpurun_out/${T}_bench_n1.json > gpurun_out/${T}_hbm_rate_dino_vitb16.txt
fi
if [ $PART = all ] || [ $PART = b ]; then
python bench.py --arch dino_vitb8 --img 96 --batch 128 --no-dp-proxy --no-fp32-rate > gpurun_out/${T}_bench_n1_dino_vitb8_96_bs128.json 2>/dev/null
python bench.py --arch dino_vits16 --img 32 --batch 128 --no-dp-proxy --no-fp32-rate > gpurun_out/${T}_bench_n1_dino_vits16_32_bs128.json 2>/dev/null
for cfg in "cait_S24_224 256" "swin_tiny_patch4_window7_224 256"; do
  set -- $cfg
  tools/prof.sh ${T}_$1_bs$2 --arch $1 --batch $2 --steps 3 --warmup 1 --graph off $Q
  python bench.py --arch $1 --batch $2 --no-dp-proxy --no-fp32-rate > gpurun_out/${T}_bench_n1_$1_bs$2.json 2>/dev/null
done
# whole-step HBM rate of the two configurations judged on HBM: PMC traffic / step time
for cfg in "cait_S24_224 256" "swin_tiny_patch4_window7_224 256"; do
  set -- $cfg
  tools/pmc.sh ${T}_$1 --arch $1 --batch $2 --steps 3 --warmup 1 --graph off $Q
  python tools/hbm_rate.py gpurun_out/${T}_$1_pmc_traffic.json gpurun_out/${T}_bench_n1_$1_bs$2.json > gpurun_out/${T}_hbm_rate_$1.txt
done
fi
if [ $PART = all ] || [ $PART = c ]; then
for a in dino_vitb16 swin_tiny_patch4_window7_224 cait_S24_224; do
  python bench.py --arch $a --mode lineareval --no-dp-proxy > gpurun_out/${T}_bench_n1_lineareval_$a.json 2>/dev/null
done
rm -f gpurun_out/${T}_shapes.txt
tools/pmc_gemm_shapes.sh ${T}
tools/clock_check.sh ${T}_vitb 32
# CaiT: the fused talking-heads attention against the three-call form, one box, interleaved twice
: > gpurun_out/${T}_ab_cait_fused_attention.txt
for rnd in 1 2; do
  for f in 0 1; do
    VITMI_TH_FUSED=$f python bench.py --arch cait_S24_224 $Q --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('VITMI_TH_FUSED=$f', 'round $rnd', d['value'], 'images/s', d['ms_per_step'], 'ms/step')
" >> gpurun_out/${T}_ab_cait_fused_attention.txt
  done
done
fi
python tools/bench_print.py gpurun_out/${T}_bench_n1*.json
