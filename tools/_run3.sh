mkdir -p gpurun_out
(timeout 1800 python -m pytest tests -m gpu -x -q) > gpurun_out/r4_gputest6.log 2>&1; tail -3 gpurun_out/r4_gputest6.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/collect_profiles.sh > gpurun_out/collect_r04.log 2>&1; tail -2 gpurun_out/collect_r04.log
