#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
for T in 32 64 128 192 256 384 512 1024; do
  echo "T=$T"
  T=$T WL=cfg4_mistral_gate_coo_k128,cfg3_mistral_down_block32_k1536,cfg4_mistral_down_coo_k128 timeout -k 10 120 python -u tools/fold_time.py 0x40 2>/dev/null || exit 1
  T=$T WL=cfg4_mistral_gate_coo_k128,cfg3_mistral_down_block32_k1536,cfg4_mistral_down_coo_k128 timeout -k 10 120 python -u tools/fold_time.py 0x0 2>/dev/null || exit 1
  T=$T WL=cfg4_mistral_gate_coo_k128,cfg3_mistral_down_block32_k1536,cfg4_mistral_down_coo_k128 timeout -k 10 120 python -u tools/fold_time.py 0x20 2>/dev/null || exit 1
done
