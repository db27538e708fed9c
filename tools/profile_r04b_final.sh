#!/bin/bash
# Round-4 (second session) evidence set, run on the GPU box through gpurun; the summaries are copied to profiles/r04_* by hand:
#   1. the bench line with the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5)
#   2. rocprofv3 --kernel-trace --stats of `bench.py --roofline-only`   (kernel averages behind the roofline legs)
#   3. kernel trace of the replayed graph, per kernel and per launch grid (tools/prof_steady.py)
#   4. configs 2 / 3 / 5 lines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4final
mkdir -p $out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench.err || exit 1
tail -c 400 $out/bench_line.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/roofline -- python3 bench.py --roofline-only > $out/roofline_leg.json 2> $out/roofline_leg.err || exit 1
cp $out/roofline/*/*kernel_stats.csv $out/roofline_leg_kernel_stats.csv
rm -rf $out/roofline
rocprofv3 --kernel-trace --output-format csv -d $out/replay -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > $out/replay_bench.json 2> $out/replay.err || exit 1
python3 tools/prof_steady.py $out/replay 1 120 "chunk_|scan_fwd_stream|gemm_|morph_|nf_|conv1d_|conv3x3|conv_s2|mamba_small|tri_|stem7" > $out/graph_replay_summary.txt 2>&1
rm -rf $out/replay
head -3 $out/graph_replay_summary.txt
python3 bench.py --infer --no-cpu-baseline --no-roofline > $out/c2.json 2>/dev/null
python3 bench.py --dtype bf16 --batch 16 --no-cpu-baseline --no-roofline > $out/c3.json 2>/dev/null
python3 bench.py --dtype bf16 --batch 2 --size 1024 --d-state 64 --steps 3 --no-cpu-baseline --no-roofline > $out/c5.json 2>/dev/null
cat $out/c2.json $out/c3.json $out/c5.json | cut -c1-220
