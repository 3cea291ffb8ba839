export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r2y; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()"
