#!/bin/bash
# usage: tools/dbg/build_variant.sh <name> <file.hip> <flags...>  -> tools/_abl/libmmunet_<name>.so: the product library with
# ONE translation unit recompiled with extra flags (A/B runs of kernel variants; select it with MMUNET_HIP_LIB)
set -euo pipefail
name=$1; file=$2; shift 2
cd "$(dirname "$0")/../../mm-unet_amd/csrc"
mkdir -p ../../tools/_abl/$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -Wno-unused-function $*"
[ "$file" = selective_scan_bwd_w8.hip ] && FLAGS="$FLAGS -fno-slp-vectorize"
objs=()
for o in *.o; do
  if [ "$o" = "${file%.hip}.o" ]; then /opt/rocm/bin/hipcc $FLAGS -c "$file" -o "../../tools/_abl/$name/$o"; objs+=("../../tools/_abl/$name/$o"); else objs+=("$o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_abl/libmmunet_$name.so "${objs[@]}"
echo built tools/_abl/libmmunet_$name.so
