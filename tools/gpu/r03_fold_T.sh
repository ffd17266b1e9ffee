#!/bin/bash
# crossover by token count: folded contraction (0x40) against the default dispatch with the fold excluded (0x80 = SDH_FWD_NO_MFMA)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
W=cfg4_mistral_gate_coo_k128,cfg4_mistral_gate_block16_k128,cfg4_mistral_down_coo_k128,cfg4_mistral_down_block16_k128,cfg3_mistral_down_block32_k1536
for T in 32 64 96 128 192 256 384 512 1024; do
  echo "T=$T fold:   $(T=$T WL=$W timeout -k 10 120 python -u tools/fold_time.py 0x40 2>/dev/null)" || exit 1
  echo "T=$T others: $(T=$T WL=$W timeout -k 10 120 python -u tools/fold_time.py 0x80 2>/dev/null)" || exit 1
done
