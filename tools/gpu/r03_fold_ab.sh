#!/bin/bash
# A/B builds of the fold kernel (tools/build_variants.sh SRC=sdh_w0fold): call time per workload
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
WL=${WL:-cfg4_mistral_gate_coo_k128,cfg4_mistral_down_coo_k128,cfg3_mistral_down_block32_k1536}
WL=$WL timeout -k 10 120 python -u tools/fold_time.py 2>/dev/null || exit 1
for v in ${VARIANTS}; do
  WL=$WL SEEDDELTA_HIP_LIB=$R/tools/bin/variants/$v/libseeddelta_hip.so timeout -k 10 120 python -u tools/fold_time.py 2>/dev/null || exit 1
done
