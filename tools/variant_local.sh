#!/bin/bash
# Builds variants of libgpc_hip.so HERE (hipcc cross-compiles without a GPU) into tools/variants/ (travels to the GPU
# box with the snapshot; *.so is git-ignored), several at a time.  usage: bash tools/variant_local.sh "name:-DFLAG ..." ...
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$R/tools/variants"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  [ "$flags" = "$spec" ] && flags=""
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags -o "$R/tools/variants/libgpc_$name.so" "$R/opengpc_amd/csrc/gpc_hip.hip" 2> "$R/tools/variants/$name.log" && echo "built $name" || echo "$name: build FAILED" ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
