mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/time_big.py product norev product norev
python tools/time_variants.py product norev
