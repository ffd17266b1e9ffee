#!/bin/bash
# dev: the persistent FFN stack with in-kernel stamps (-DSDH_STK_STAMPS) into tools/bin/variants/stkstamps/
set -e
cd "$(dirname "$0")/../llama.cpp_amd/csrc"
make -j8 2>&1 | grep -E "error|warning" || true
out=../../tools/bin/variants/stkstamps; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DSDH_STK_STAMPS -c sdh_stack.hip -o $out/sdh_stack.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libseeddelta_hip.so build/sdh_api.o build/sdh_base.o build/sdh_delta.o build/sdh_decode.o build/sdh_scan.o build/sdh_spmm.o build/sdh_w0mfma.o $out/sdh_stack.o
echo built stkstamps
