#!/bin/bash
# The failing invocation of round 5's cooperative-launch experiment, for the record (profiles/r06_coop.txt): two contexts of one process on two
# host threads, their persistent launches issued with hipLaunchCooperativeKernel (TPOSE_COOPERATIVE=1, tp_persist.hip: launch_one) instead of
# plain launches.  Plain launches (the product) pass the same script.  Runs on the GPU box.
cd "${GRAFT_REPO_ROOT:-.}"
echo "== TPOSE_COOPERATIVE=0 (the product's plain launches)"
timeout 300 python tools/two_contexts.py 2>&1 | tail -3; echo "exit code $?"
echo "== TPOSE_COOPERATIVE=1, python -X faulthandler"
TPOSE_COOPERATIVE=1 timeout 300 python -X faulthandler tools/two_contexts.py 2>&1 | tail -40; echo "exit code ${PIPESTATUS[0]}"
G=$(which rocgdb 2>/dev/null || which gdb 2>/dev/null || ls /opt/rocm/bin/rocgdb 2>/dev/null)
if [ -n "$G" ]; then
  echo "== TPOSE_COOPERATIVE=1 under $G"
  TPOSE_COOPERATIVE=1 timeout 600 $G -batch -ex "set pagination off" -ex run -ex "thread apply all bt 25" --args python tools/two_contexts.py 2>&1 | tail -120
else
  echo "(no gdb / rocgdb in the image: the Python-level trace above is what there is)"
fi
