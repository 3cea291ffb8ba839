export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r2z; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in pad0 pad1 pad2; do TPOSE_HIP_LIB=$PWD/tpose_amd/variants/libtpose_hip_$v.so python tools/time_acc.py >> $O/ab.jsonl 2>$O/ab_$v.err; done
cat $O/ab.jsonl | cut -c1-190
for v in pad1 pad2; do TPOSE_HIP_LIB=$PWD/tpose_amd/variants/libtpose_hip_$v.so timeout 300 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "fused or piecewise or golden or soak" 2>&1 | tail -2; done
