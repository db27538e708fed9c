#!/bin/bash
# A/B of the streaming forward at the headline shape by MMU_SCAN_STREAM version (1 = two-barrier kernel of rounds 2-3,
# 2 = one-barrier variant, 3 = 16 tokens per lane).  usage (GPU box): tools/ab_scan_fwd.sh <tag> [versions...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
tag=${1:-x}; shift
for v in ${@:-1 3}; do MMU_SCAN_STREAM=$v bash tools/time_scan_fwd.sh "${tag}_v$v" 20; done
