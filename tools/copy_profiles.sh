#!/bin/bash
# gpurun_out/r05 (what tools/collect_profiles.sh left on the GPU box) -> profiles/r05_* (tracked)
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r05; D=profiles
cp $S/bench.json $D/r05_bench.json
cp $S/bench_20_5.json $D/r05_bench_steps20_warmup5.json
cp $S/kt/kt_kernel_stats.csv $D/r05_kernel_stats.csv
cp $S/kt20/kt_kernel_stats.csv $D/r05_kernel_stats_steps20_warmup5.csv
cp $S/persist_timeline.json $D/r05_persist_timeline.json
cp $S/persist_timeline_4000.json $D/r05_persist_timeline_after4000.json
cp $S/persist_check.txt $D/r05_persist_check.txt
cp $S/long_parity.txt $D/r05_long_parity.txt
cp $S/long_run_timing.txt $D/r05_long_run_timing.txt
cp $S/ipc_handover.txt $D/r05_ipc_handover.txt
cp $S/band_timing_4096_12000.json $D/r05_band_timing_4096_12000.json
cp $S/band_timing_2048_3000.json $D/r05_band_timing_2048_3000.json
cp $S/config2.txt $D/r05_config2_schedule.txt
cp $S/config3.txt $D/r05_config3_warp.txt
cp $S/config4.json $D/r05_config4_batch.json
[ -f $S/wave_timeline.json ] && cp $S/wave_timeline.json $D/r05_wave_timeline.json
[ -f $S/time_4096.txt ] && cp $S/time_4096.txt $D/r05_time_4096.txt
[ -f $S/bench_two_ranks_one_gpu.json ] && grep '^{' $S/bench_two_ranks_one_gpu.json > $D/r05_bench_two_ranks_one_gpu.json
[ -f $S/call_length.txt ] && cp $S/call_length.txt $D/r05_call_length.txt
[ -f $S/pmc_persist.json ] && cp $S/pmc_persist.json $D/r05_pmc_persist.json
[ -f $S/launch_profile.txt ] && cp $S/launch_profile.txt $D/r05_launch_profile.txt
[ -f $S/pmc_traffic_4096_12000.json ] && cp $S/pmc_traffic_4096_12000.json $D/r05_pmc_traffic_4096_12000.json
[ -f $S/pmc_traffic_2048_3000.json ] && cp $S/pmc_traffic_2048_3000.json $D/r05_pmc_traffic_2048_3000.json
ls -la $D/r05_*
[ -f $S/contrast_sweep.txt ] && cp $S/contrast_sweep.txt $D/r05_contrast_sweep.txt
