#!/bin/bash
# Builds libmmunet_hip.so for gfx950 in-tree (cross-compiles without a GPU).
#
# Incremental by CONTENT: an object is rebuilt when the sha256 of (its source, the shared headers, the flags) differs
# from the one recorded beside it (<name>.o.hash) -- not by mtime, which a checkout or a snapshot copy resets.
# MMU_FORCE_REBUILD=1 compiles everything regardless (what to use to show that the tree compiles from scratch).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -Wall -Wno-unused-function ${MMU_EXTRA_FLAGS:-}"
# left-overs of -save-temps runs (tools/kstats.sh and friends): never part of the library, 30 MB per gpurun push
rm -f ./*-hip-amdgcn-amd-amdhsa-gfx950.* ./*-host-x86_64-unknown-linux-gnu.* 2>/dev/null || true
SOURCES="mmu_abi.hip selective_scan.hip selective_scan_stream.hip selective_scan_bwd_w8.hip causal_conv1d.hip morph_sample.hip morph_coords.hip resize.hip conv3x3_small.hip tri_order.hip norm_fused.hip mamba_pre.hip conv3x3_mfma.hip conv3x3_wgrad_mfma.hip gemm_tokens_mfma.hip gemm_nt_splitk.hip pointwise_one.hip sum_parts.hip maxpool.hip conv7x7_small.hip gated_mul.hip cbam_stats.hip mamba_small_fused.hip conv_s2_mfma.hip morph_mix.hip adamw_multi.hip deferred_reduce.hip dice_bce.hip dt_proj.hip tri_fused.hip stem7_mfma.hip offset_conv_mfma.hip"
objs=()
compiled=0
for f in $SOURCES; do
  o="${f%.hip}.o"
  extra=""
  # the SLP vectorizer re-packs the per-token sums of the w8 backward with register shuffles (3 v_mov per pair)
  [ "$f" = selective_scan_bwd_w8.hip ] && extra="-fno-slp-vectorize"
  want=$( (cat "$f" mmu_common.h scan_common.h ../../include/mmunet_amd.h; echo "$FLAGS $extra") | sha256sum | cut -d' ' -f1)
  have=$(cat "$o.hash" 2>/dev/null || true)
  if [ "${MMU_FORCE_REBUILD:-0}" = 1 ] || [ ! -f "$o" ] || [ "$want" != "$have" ]; then
    rm -f "$o.hash"
    ( $HIPCC $FLAGS $extra -c "$f" -o "$o" && echo "$want" > "$o.hash" ) &
    compiled=$((compiled + 1))
  fi
  objs+=("$o")
done
wait
for o in "${objs[@]}"; do
  [ -f "$o.hash" ] || { echo "build.sh: compiling ${o%.o}.hip failed" >&2; exit 1; }
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libmmunet_hip.so "${objs[@]}"
echo "built $(pwd)/libmmunet_hip.so ($compiled of ${#objs[@]} objects compiled)"
