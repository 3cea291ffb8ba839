#!/bin/bash
# gpurun_out/r03 (what tools/collect_profiles.sh left on the GPU box) -> profiles/r03_* (tracked)
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r03; D=profiles
cp $S/bench.json $D/r03_bench.json
cp $S/bench_20_5.json $D/r03_bench_steps20_warmup5.json
cp $S/kt/kt_kernel_stats.csv $D/r03_kernel_stats.csv
cp $S/kt20/kt_kernel_stats.csv $D/r03_kernel_stats_steps20_warmup5.csv
cp $S/persist_timeline.json $D/r03_persist_timeline.json
cp $S/persist_timeline_4000.json $D/r03_persist_timeline_after4000.json
cp $S/persist_check.txt $D/r03_persist_check.txt
cp $S/long_parity.txt $D/r03_long_parity.txt
cp $S/long_run_timing.txt $D/r03_long_run_timing.txt
cp $S/ipc_handover.txt $D/r03_ipc_handover.txt
cp $S/band_timing_4096_12000.json $D/r03_band_timing_4096_12000.json
cp $S/band_timing_2048_3000.json $D/r03_band_timing_2048_3000.json
cp $S/config2.txt $D/r03_config2_schedule.txt
cp $S/config3.txt $D/r03_config3_warp.txt
cp $S/config4.json $D/r03_config4_batch.json
[ -f $S/call_length.txt ] && cp $S/call_length.txt $D/r03_call_length.txt
[ -f $S/pmc_persist.json ] && cp $S/pmc_persist.json $D/r03_pmc_persist.json
ls -la $D/r03_*
