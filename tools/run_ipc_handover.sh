#!/bin/bash
# Row e3: device-side hand-over between two processes through an IPC-mapped mailbox (tools/ipc_handover.cpp), both on device 0.
# Prints the JSON lines of rank a (the side that times the round trips).  Needs an MI355X and HSA_ENABLE_IPC_MODE_LEGACY=0.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
D=$(mktemp -d)
hipcc --offload-arch=gfx950 -O2 -o $D/ipc_handover tools/ipc_handover.cpp 2>/dev/null
timeout 120 $D/ipc_handover b $D/handle > $D/b.txt 2>&1 &
PB=$!
timeout 120 $D/ipc_handover a $D/handle > $D/a.txt 2>&1 || true
wait $PB || true
cat $D/a.txt
echo "--- rank b"; cat $D/b.txt
rm -rf $D
