mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q -k "two_ranks") > gpurun_out/r4_t8.log 2>&1; tail -15 gpurun_out/r4_t8.log
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 256 --warmup 32 --backend gloo --share-gpu > gpurun_out/r4_pair2.json 2> gpurun_out/r4_pair2.err; python -c "
import json
for l in open('gpurun_out/r4_pair2.json'):
    if l.startswith('{'): print(json.loads(l)['pair_split'])"; tail -3 gpurun_out/r4_pair2.err
