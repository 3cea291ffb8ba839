export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3b; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
python tools/time_acc.py > $O/time.json 2>$O/time.err; cat $O/time.json
