#!/bin/bash
# dev: build A/B variants of the fused decode kernel into tools/bin/variants/<name>/libseeddelta_hip.so
set -e
cd "$(dirname "$0")/../llama.cpp_amd/csrc"
make -j8 >/dev/null
for v in ${VARIANTS:-"t256:-DSDH_DECODE_L4096_T256" "t1024:-DSDH_DECODE_L4096_T1024" "swizzle:-DSDH_BASE_SWIZZLE"}; do
  name=${v%%:*}; flag=${v#*:}
  out=../../tools/bin/variants/$name; mkdir -p $out
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flag -c sdh_decode.hip -o $out/sdh_decode.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libseeddelta_hip.so build/sdh_api.o build/sdh_base.o build/sdh_delta.o $out/sdh_decode.o build/sdh_scan.o build/sdh_spmm.o build/sdh_w0mfma.o
  echo built $name
done
