#!/bin/bash
# Diagnostic build of the library with s_memtime stamps in the streaming scan (never the product build):
# builds tools/_abl/libmmunet_stamps.so here (CPU container), run tools/stream_stamps.py on the GPU box.
set -euo pipefail
cd "$(dirname "$0")/../mm-unet_amd/csrc"
mkdir -p ../../tools/_abl/stamps
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -DMMU_STREAM_STAMPS"
objs=()
for f in mmu_abi.hip selective_scan.hip selective_scan_stream.hip selective_scan_bwd_w8.hip causal_conv1d.hip morph_sample.hip morph_coords.hip resize.hip conv3x3_small.hip tri_order.hip norm_fused.hip mamba_pre.hip conv3x3_mfma.hip conv3x3_wgrad_mfma.hip gemm_tokens_mfma.hip; do
  o="../../tools/_abl/stamps/${f%.hip}.o"
  if [ "$f" = selective_scan_stream.hip ]; then /opt/rocm/bin/hipcc $FLAGS -c "$f" -o "$o"; else o="${f%.hip}.o"; fi
  objs+=("$o")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_abl/libmmunet_stamps.so "${objs[@]}"
echo built tools/_abl/libmmunet_stamps.so
