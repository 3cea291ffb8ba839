#!/bin/bash
# gpurun_out/r06 (what tools/collect_profiles.sh left on the GPU box) -> profiles/r06_* (tracked)
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r06; D=profiles
cpif() { if [ -f "$1" ]; then cp "$1" "$2"; fi; }
cpif $S/bench.json $D/r06_bench.json
cpif $S/bench_20_5.json $D/r06_bench_steps20_warmup5.json
cpif $S/kt/kt_kernel_stats.csv $D/r06_kernel_stats.csv
cpif $S/kt20/kt_kernel_stats.csv $D/r06_kernel_stats_steps20_warmup5.csv
cpif $S/keep/kernel_stats_256.csv $D/r06_kernel_stats_256.csv
cpif $S/keep/kernel_trace_k_persist_256.csv $D/r06_kernel_trace_k_persist_256.csv
cpif $S/persist_timeline.json $D/r06_persist_timeline.json
cpif $S/persist_timeline_4000.json $D/r06_persist_timeline_after4000.json
cpif $S/persist_timeline_synthetic.json $D/r06_persist_timeline_synthetic_x0.10.json
cpif $S/wave_timeline.json $D/r06_wave_timeline.json
cpif $S/launch_profile.txt $D/r06_launch_profile.txt
cpif $S/persist_check.txt $D/r06_persist_check.txt
cpif $S/time_4096.txt $D/r06_time_4096.txt
cpif $S/long_parity.txt $D/r06_long_parity.txt
cpif $S/long_run_timing.txt $D/r06_long_run_timing.txt
cpif $S/call_length.txt $D/r06_call_length.txt
cpif $S/contrast_sweep.txt $D/r06_contrast_sweep.txt
cpif $S/photo_timing.txt $D/r06_photo_timing.txt
cpif $S/ta_bench.txt $D/r06_ta_bench.txt
cpif $S/pmc_traffic_4096_12000.json $D/r06_pmc_traffic_4096_12000.json
cpif $S/pmc_traffic_2048_3000.json $D/r06_pmc_traffic_2048_3000.json
cpif $S/pmc_traffic_2048_3000_synthetic.json $D/r06_pmc_traffic_2048_3000_synthetic_x0.10.json
cpif $S/pmc_persist.json $D/r06_pmc_persist.json
cpif $S/ipc_handover.txt $D/r06_ipc_handover.txt
cpif $S/band_timing_4096_12000.json $D/r06_band_timing_4096_12000.json
cpif $S/band_timing_2048_3000.json $D/r06_band_timing_2048_3000.json
cpif $S/config2.txt $D/r06_config2_schedule.txt
cpif $S/config2_meninas.txt $D/r06_config2_meninas.txt
cpif $S/config3.txt $D/r06_config3_warp.txt
cpif $S/config4.json $D/r06_config4_batch.json
cpif $S/coop.txt $D/r06_coop.txt
cpif $S/long_mixed_calls_meninas.txt $D/r06_long_mixed_calls_meninas.txt
if [ -f $S/bench_two_ranks_one_gpu.json ]; then grep '^{' $S/bench_two_ranks_one_gpu.json > $D/r06_bench_two_ranks_one_gpu.json || true; fi
ls -la $D/r06_*
