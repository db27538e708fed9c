#!/bin/bash
# Builds tools/_abl/libmmunet_smallstamps.so = the library with csrc/mamba_small_fused.hip compiled -DMMU_SMALL_STAMPS
# (s_memtime at the phase boundaries, read by tools/dbg/small_stamps.py).  Run here (CPU container); the .so travels.
set -euo pipefail
cd /root/repo/mm-unet_amd/csrc
mkdir -p ../../tools/_abl/smallstamps
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -DMMU_SMALL_STAMPS"
/opt/rocm/bin/hipcc $FLAGS -c mamba_small_fused.hip -o ../../tools/_abl/smallstamps/mamba_small_fused.o
objs=()
for o in *.o; do
  if [ "$o" = mamba_small_fused.o ]; then objs+=(../../tools/_abl/smallstamps/mamba_small_fused.o); else objs+=("$o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_abl/libmmunet_smallstamps.so "${objs[@]}"
echo built tools/_abl/libmmunet_smallstamps.so
