#!/bin/bash
# dev: A/B variants of sdh_scan.hip into tools/bin/variants/<name>/libseeddelta_hip.so
set -e
cd "$(dirname "$0")/../llama.cpp_amd/csrc"
make -j8 >/dev/null
for v in ${VARIANTS}; do
  name=${v%%:*}; flag=${v#*:}; flag=${flag//,/ }
  out=../../tools/bin/variants/$name; mkdir -p $out
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flag -c sdh_scan.hip -o $out/sdh_scan.o
  objs=""; for o in build/*.o; do [ "$o" = "build/sdh_scan.o" ] || objs="$objs $o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libseeddelta_hip.so $objs $out/sdh_scan.o
  echo built $name
done
