#!/bin/bash
# per-kernel split of one scan-backward shape: bash tools/dbg/prof_bwd_split.sh <tag> [B D L]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-w8}; shift
out=gpurun_out/w8/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/run -- python3 tools/dbg/bwd_one.py ${@:-8 128 65536} 5 > $out/log.txt 2>&1
cp $out/run/*/*kernel_stats.csv $out/stats.csv
cut -d, -f1-4 $out/stats.csv | cut -c1-140 | head -10
