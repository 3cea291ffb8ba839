#!/bin/bash
# Runs on the GPU box (gpurun): everything profiles/r06_* is derived from, into gpurun_out/r06/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
# 1. the bench line: the driver's flags, and the default flags (5 x 2048 steps); the per-dispatch rows of the roofline's kernel kept
TPOSE_BENCH_KEEP_TRACE=$O/keep20 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
TPOSE_BENCH_KEEP_TRACE=$O/keep python bench.py > $O/bench.json 2> $O/bench.err
# 2. kernel trace of the bench command (driver's flags and default flags)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --no-cpu-baseline --no-pmc > $O/kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt20 -o kt -- python bench.py --no-cpu-baseline --no-pmc --steps 20 --warmup 5 > $O/kt20.log 2>&1
find $O -name "*kernel_trace.csv" -path "*kt*" -delete
# 3. in-kernel timeline of the persistent kernel (debug flavour of the library) on the headline picture: fresh mesh, and after 4000 grad-iters;
#    and on the synthetic raster of rounds 3-5
TPOSE_PHOTO=meninas python tools/persist_timeline.py --rebuild > $O/persist_timeline.json 2> $O/persist_timeline.err
TPOSE_PHOTO=meninas TPOSE_TIMELINE_AFTER=4000 python tools/persist_timeline.py > $O/persist_timeline_4000.json 2>> $O/persist_timeline.err
python tools/persist_timeline.py > $O/persist_timeline_synthetic.json 2>> $O/persist_timeline.err
TPOSE_WAVES=12 python tools/wave_timeline.py --rebuild > $O/wave_timeline.json 2>> $O/persist_timeline.err
python tools/launch_profile.py > $O/launch_profile.txt 2>> $O/persist_timeline.err   # the first grad-iters of a launch, one by one
# 4. persistent path against the two-kernel path and the oracle: parity and timing, other configurations
python tools/persist_check.py > $O/persist_check.txt 2>&1
python tools/time_big.py product > $O/time_4096.txt 2>&1
python tools/long_parity.py > $O/long_parity.txt 2>&1
TPOSE_PHOTO=meninas python tools/long_mixed_calls.py > $O/long_mixed_calls_meninas.txt 2>&1   # 120 000 grad-iters of the photograph in calls of mixed lengths against the two-kernel path
python tools/time_variants.py product > $O/long_run_timing.txt 2>&1
python tools/call_length.py > $O/call_length.txt 2>&1
python tools/contrast_sweep.py > $O/contrast_sweep.txt 2>&1
python tools/photo_timing.py synthetic_x0.10 meninas fruit imageA imageB shoeA shoeB synthetic_x0.30 synthetic_x1.00 > $O/photo_timing.txt 2>&1
python tools/pmc_size.py 4096 12000 > $O/pmc_traffic_4096_12000.json 2> $O/pmc_size.err   # memory-side traffic per grad-iter (separate --pmc passes)
python tools/pmc_size.py 2048 3000 > $O/pmc_traffic_2048_3000_synthetic.json 2>> $O/pmc_size.err
TPOSE_PHOTO=meninas python tools/pmc_size.py 2048 3000 > $O/pmc_traffic_2048_3000.json 2>> $O/pmc_size.err
TPOSE_PMC_TARGET=persist python tools/pmc_kernels.py $O/pmc_persist.json > /dev/null 2> $O/pmc_persist.err   # counters of k_persist (separate --pmc passes)
# 5. row e3: a hand-over between two processes through an IPC-mapped granule; the band split's protocol with both bands on this device
bash tools/run_ipc_handover.sh > $O/ipc_handover.txt 2>&1
python tools/band_timing.py > $O/band_timing_4096_12000.json 2> $O/band_timing.err
python tools/band_timing.py 2048 3000 > $O/band_timing_2048_3000.json 2>> $O/band_timing.err
HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 256 --warmup 32 --backend gloo --share-gpu > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks.err
# 6. the schedules
python tools/run_config2.py 600 > $O/config2.txt 2>&1
(echo "config 2 on resource/meninas.png itself (the decoded fixture), the reference's window 800 x 920: the first 300 000 frames of the schedule"; \
 echo "== frames in chunks on the device (tp_iterate_frames)"; CONFIG2_PHOTO=meninas python tools/run_config2.py 600 -maxframes 300000; \
 echo "== frame by frame (-nochunks: round 5's loop)"; CONFIG2_PHOTO=meninas python tools/run_config2.py 600 -maxframes 300000 -nochunks; \
 echo "== 3 000 000 frames, in chunks"; CONFIG2_PHOTO=meninas CONFIG2_VERBOSE=1 python tools/run_config2.py 900 -maxframes 3000000 | tail -12) > $O/config2_meninas.txt 2>&1
python tools/run_config3.py > $O/config3.txt 2>&1
python tools/run_batch.py --pairs 8 --iters 512 > $O/config4.json 2> $O/config4.err
ls -la $O
# then, in the development container: tools/copy_profiles.sh (gpurun_out/r06 -> profiles/r06_*)
