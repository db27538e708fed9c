#!/bin/bash
# Kernel trace of the replayed training step only (tools/prof_steady.py summary) -> gpurun_out/$1/graph_replay_summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-replay}
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/replay -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > $out/replay_bench.json 2> $out/replay.err || exit 1
python3 tools/prof_steady.py $out/replay 1 120 "chunk_|scan_fwd_stream|gemm_|morph_|nf_|conv1d_|conv3x3|conv_s2|mamba_small|tri_" > $out/graph_replay_summary.txt 2>&1
rm -rf $out/replay
head -3 $out/graph_replay_summary.txt
