#!/bin/bash
# two ranks on one GPU over gloo: rehearsal of the multi-rank bench path (the driver runs the real thing on 8 GPUs)
set -o pipefail
mkdir -p gpurun_out
SDH_BENCH_DETAIL=$GRAFT_REPO_ROOT/gpurun_out/r03_rehearse_2ranks_detail.json SDH_BENCH_ONE_DEVICE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 64 --warmup 8 > gpurun_out/r03_rehearse_2ranks.log 2>&1 || { tail -20 gpurun_out/r03_rehearse_2ranks.log; exit 1; }
grep '^{' gpurun_out/r03_rehearse_2ranks.log | tail -1 > gpurun_out/r03_rehearse_2ranks_line.json
cat gpurun_out/r03_rehearse_2ranks_line.json
