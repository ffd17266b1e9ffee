#!/bin/bash
R=$GRAFT_REPO_ROOT
run() { env "$@" timeout -k 10 120 python $R/tools/stack_ab.py 2>&1 | grep "us/layer" >> $R/gpurun_out/stack_ab.txt || exit 1; }
: > $R/gpurun_out/stack_ab.txt
timeout -k 10 200 python -u -m pytest $R/tests/test_gpu_stack.py -x -q --timeout 100 --timeout-method thread 2>&1 | tail -2
for cfg in "$@"; do run $cfg || exit 1; done
cat $R/gpurun_out/stack_ab.txt
