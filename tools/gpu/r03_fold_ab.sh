#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
WL=${WL:-cfg4_mistral_gate_coo_k128,cfg4_mistral_down_coo_k128,cfg3_mistral_down_block32_k1536}
WL=$WL timeout -k 10 120 python -u tools/fold_time.py 2>/dev/null || exit 1
for v in ${VARIANTS}; do
  WL=$WL SEEDDELTA_HIP_LIB=$R/tools/bin/variants/$v/libseeddelta_hip.so timeout -k 10 120 python -u tools/fold_time.py 2>/dev/null || exit 1
  WL=$WL SDH_FOLD_WAVES=8 SEEDDELTA_HIP_LIB=$R/tools/bin/variants/$v/libseeddelta_hip.so timeout -k 10 120 python -u tools/fold_time.py 2>/dev/null || exit 1
done
