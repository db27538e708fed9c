#!/bin/bash
# Round-4 final evidence set (third session), run on the GPU box through gpurun; the summaries are copied to profiles/r04_* by
# hand.  = tools/profile_r04b_final.sh + the counter passes behind every `traffic` field of the bench line:
#   scan forward / backward (tools/pmc_traffic.sh), conv3x3_mfma, gemm_nt (tools/pmc_traffic_generic.sh), and the
#   per-kernel HBM bytes of one eager step (tools/pmc_step_traffic.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4final
mkdir -p $out
bash tools/pmc_traffic.sh fwd > $out/scan_fwd_traffic.log 2>&1 || exit 1
bash tools/pmc_traffic.sh bwd > $out/scan_bwd_traffic.log 2>&1 || exit 1
cp profiles/scan_fwd_traffic.json profiles/scan_bwd_traffic.json $out/
bash tools/pmc_traffic_generic.sh conv3 conv3x3_mfma tools/prof_conv3x3_mfma_only.py 3 > $out/conv3x3_traffic.txt 2>&1 || exit 1
bash tools/pmc_traffic_generic.sh gemmnt gemm_nt tools/prof_gemm_nt_only.py > $out/gemm_nt_traffic.txt 2>&1 || exit 1
bash tools/pmc_step_traffic.sh > $out/step_traffic_per_kernel.txt 2>&1 || exit 1
echo "traffic passes done"
bash tools/profile_r04b_final.sh
