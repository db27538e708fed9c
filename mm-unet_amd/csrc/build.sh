#!/bin/bash
# Builds libmmunet_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -Wall -Wno-unused-function ${MMU_EXTRA_FLAGS:-}"
objs=()
for f in mmu_abi.hip selective_scan.hip selective_scan_stream.hip selective_scan_bwd_w8.hip causal_conv1d.hip morph_sample.hip morph_coords.hip resize.hip conv3x3_small.hip tri_order.hip norm_fused.hip mamba_pre.hip conv3x3_mfma.hip conv3x3_wgrad_mfma.hip gemm_tokens_mfma.hip gemm_nt_splitk.hip pointwise_one.hip sum_parts.hip maxpool.hip conv7x7_small.hip gated_mul.hip cbam_stats.hip; do
  o="${f%.hip}.o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ mmu_common.h -nt "$o" ] || [ scan_common.h -nt "$o" ] || [ ../../include/mmunet_amd.h -nt "$o" ]; then
    extra=""
    # the SLP vectorizer re-packs the per-token sums of the w8 backward with register shuffles (3 v_mov per pair)
    [ "$f" = selective_scan_bwd_w8.hip ] && extra="-fno-slp-vectorize"
    $HIPCC $FLAGS $extra -c "$f" -o "$o" &
  fi
  objs+=("$o")
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libmmunet_hip.so "${objs[@]}"
echo "built $(pwd)/libmmunet_hip.so"
