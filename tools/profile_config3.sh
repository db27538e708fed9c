#!/bin/bash
# Kernel trace of the replayed bf16 training step of BASELINE config 3 (bs 16) -> gpurun_out/$1/config3_summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-c3}
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/replay -- python3 bench.py --dtype bf16 --batch 16 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $out/c3_bench.json 2> $out/c3.err || exit 1
python3 tools/prof_steady.py $out/replay 1 120 "chunk_|scan_fwd_stream|gemm_|morph_|nf_|conv1d_|conv3x3|conv_s2|mamba_small|tri_" > $out/config3_summary.txt 2>&1
rm -rf $out/replay
head -3 $out/config3_summary.txt
