#!/bin/bash
# two ranks on one GPU over gloo: rehearsal of the multi-rank bench path (the driver runs the real thing on 8 GPUs)
set -o pipefail
mkdir -p gpurun_out
SDH_BENCH_ONE_DEVICE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 512 --warmup 64 > gpurun_out/bench_rehearse.log 2>&1 || { tail -20 gpurun_out/bench_rehearse.log; exit 1; }
python -c "import json;d=json.loads([l for l in open('gpurun_out/bench_rehearse.log').read().strip().splitlines() if l.startswith('{')][-1]);print(d['value'], d['n_gpus'], d['scan']['value'], d['scan']['n_gpus'], d['scan']['candidates'])"
