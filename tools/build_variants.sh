#!/bin/bash
# dev: build A/B variants of one translation unit into tools/bin/variants/<name>/libseeddelta_hip.so
#   SRC=sdh_decode VARIANTS="name:-DFLAG,-DFLAG2 ..." tools/build_variants.sh
set -e
cd "$(dirname "$0")/../llama.cpp_amd/csrc"
make -j8 >/dev/null
SRC=${SRC:-sdh_decode}
for v in ${VARIANTS}; do
  name=${v%%:*}; flag=${v#*:}; flag=${flag//,/ }
  out=../../tools/bin/variants/$name; mkdir -p $out
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flag -c $SRC.hip -o $out/$SRC.o
  objs=""; for o in build/*.o; do [ "$o" = "build/$SRC.o" ] || objs="$objs $o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libseeddelta_hip.so $objs $out/$SRC.o
  echo built $name
done
