#!/bin/bash
# usage (here, after tools/refresh_profiles.sh ran on the GPU box): tools/collect_profiles.sh [tag]
# copies the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked)
set +e
T=${1:-r04}
G=gpurun_out
P=profiles
cp $G/${T}_bench_n1*.json $P/
cp $G/${T}_vitb.txt $P/${T}_rocprofv3_summary.txt
cp $G/${T}_vitb_kernel_stats.csv $P/${T}_rocprofv3_kernel_stats_bench_steps5.csv
cp $G/${T}_vitb.json $P/${T}_bench_under_rocprof.json
cp $G/${T}_vitb_buckets.json $P/${T}_rocprof_buckets.json
cp $G/${T}_vitb_serial.txt $P/${T}_rocprofv3_summary_serial.txt
cp $G/${T}_vitb_serial_kernel_stats.csv $P/${T}_rocprofv3_kernel_stats_bench_steps5_serial.csv
cp $G/${T}_vitb_pmc_traffic.json $P/${T}_pmc_traffic.json
cp $G/${T}_vitb_pmc_traffic.txt $P/${T}_pmc_traffic.txt
cp $G/${T}_vitb_pmc_mfma.json $P/${T}_pmc_mfma_busy.json
cp $G/${T}_vitb_pmc_mfma.txt $P/${T}_pmc_mfma_busy.txt
for c in cait_S24_224_bs256 swin_tiny_patch4_window7_224_bs256; do
  cp $G/${T}_$c.txt $P/${T}_rocprofv3_summary_$c.txt
  cp $G/${T}_${c}_kernel_stats.csv $P/${T}_rocprofv3_kernel_stats_$c.csv
done
cp $G/${T}_vitb_clock.txt $P/${T}_clock_two_methods.txt
cp $G/${T}_hbm_rate_*.txt $P/
cp $G/${T}_ab_*.txt $P/
for a in cait_S24_224 swin_tiny_patch4_window7_224; do cp $G/${T}_${a}_pmc_traffic.txt $P/${T}_pmc_traffic_$a.txt; done
cp $G/${T}_gemm_traffic_by_shape.txt $G/${T}_gemm_traffic_by_shape.json $P/
ls $P | grep "^${T}_"
