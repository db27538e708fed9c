#!/bin/bash
# usage: build_variant.sh <name> <flags...>  -> tools/_abl/libmmunet_<name>.so (w8 file compiled with the flags)
set -euo pipefail
name=$1; shift
cd /root/repo/mm-unet_amd/csrc
mkdir -p ../../tools/_abl/$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -fno-slp-vectorize $*"
objs=()
for f in mmu_abi.hip selective_scan.hip selective_scan_stream.hip selective_scan_bwd_w8.hip causal_conv1d.hip morph_sample.hip morph_coords.hip resize.hip conv3x3_small.hip tri_order.hip norm_fused.hip mamba_pre.hip conv3x3_mfma.hip conv3x3_wgrad_mfma.hip gemm_tokens_mfma.hip; do
  o="../../tools/_abl/$name/${f%.hip}.o"
  if [ "$f" = selective_scan_bwd_w8.hip ]; then /opt/rocm/bin/hipcc $FLAGS -c "$f" -o "$o"; else o="${f%.hip}.o"; fi
  objs+=("$o")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_abl/libmmunet_$name.so "${objs[@]}"
echo built tools/_abl/libmmunet_$name.so
