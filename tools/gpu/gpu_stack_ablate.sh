#!/bin/bash
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/stack_ablate.txt
for a in 0 1 2 4 8 16 32 7 24 31 63; do
  SDH_STACK_ABLATE=$a timeout -k 10 120 python $R/tools/stack_ab.py 2>&1 | grep "us/layer" >> $R/gpurun_out/stack_ablate.txt || exit 1
done
cat $R/gpurun_out/stack_ablate.txt
