export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r2t; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_configs.py tests/test_harness.py tests/test_distributed.py -q -m gpu -x 2>&1 | tail -25 > $O/pytest.txt; cat $O/pytest.txt
