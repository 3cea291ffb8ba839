mkdir -p gpurun_out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python tools/pmc_size.py 4096 12000 > gpurun_out/r4_pmc_4096.json 2>gpurun_out/r4_pmc_4096.err; cat gpurun_out/r4_pmc_4096.json; tail -3 gpurun_out/r4_pmc_4096.err
timeout 600 python tools/pmc_size.py 2048 3000 > gpurun_out/r4_pmc_2048.json 2>gpurun_out/r4_pmc_2048.err; cat gpurun_out/r4_pmc_2048.json
