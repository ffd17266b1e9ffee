#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 200 python -u tools/select_time.py 2>/dev/null || exit 1
for v in ${VARIANTS}; do
  SEEDDELTA_HIP_LIB=$R/tools/bin/variants/$v/libseeddelta_hip.so timeout -k 10 200 python -u tools/select_time.py 2>/dev/null || exit 1
done
