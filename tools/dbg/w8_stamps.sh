#!/bin/bash
# Diagnostic build of the library with s_memtime stamps in the w8 scan backward (never the product build):
# builds tools/_abl/libmmunet_w8stamps.so here (CPU container); run tools/dbg/w8_stamps.py on the GPU box.
set -euo pipefail
cd "$(dirname "$0")/../../mm-unet_amd/csrc"
mkdir -p ../../tools/_abl/w8stamps
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -fno-slp-vectorize -DMMU_W8_STAMPS ${W8_EXTRA:-}"
objs=()
for f in $(ls *.hip); do
  o="../../tools/_abl/w8stamps/${f%.hip}.o"
  if [ "$f" = selective_scan_bwd_w8.hip ]; then /opt/rocm/bin/hipcc $FLAGS -c "$f" -o "$o"; else o="${f%.hip}.o"; fi
  objs+=("$o")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_abl/libmmunet_w8stamps.so "${objs[@]}"
echo built tools/_abl/libmmunet_w8stamps.so
