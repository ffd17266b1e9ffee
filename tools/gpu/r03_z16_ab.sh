#!/bin/bash
# pre-pass variants: kernel time of fold_z16_kernel from rocprofv3 kernel stats
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in default ${VARIANTS}; do
  for wl in cfg4_mistral_gate_coo_k128 cfg4_mistral_down_coo_k128; do
    rm -rf $R/gpurun_out/z16_$v
    if [ $v = default ]; then unset SEEDDELTA_HIP_LIB; else export SEEDDELTA_HIP_LIB=$R/tools/bin/variants/$v/libseeddelta_hip.so; fi
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/z16_$v -- python3 $R/tools/fold_prof.py $wl 40 > $R/gpurun_out/z16_$v.log 2>&1 || { tail -5 $R/gpurun_out/z16_$v.log; exit 1; }
    f=$(find $R/gpurun_out/z16_$v -name "*kernel_stats.csv" | head -1)
    echo "$v $wl $(grep z16 $f | cut -d, -f2-4)"
    rm -rf $R/gpurun_out/z16_$v
  done
done
