#!/bin/bash
# Round-4 evidence set (run on the GPU box through gpurun); the summaries are copied to profiles/r04_* by hand:
#   1. the bench line with the command the driver used in round 2 (python bench.py --gpus 1 --steps 20 --warmup 5)
#   2. rocprofv3 --kernel-trace --stats of `bench.py --roofline-only`   (kernel averages behind the roofline legs)
#   3. HBM traffic of one scan forward / backward call (tools/pmc_traffic.sh: separate --pmc passes)
#   4. kernel trace of the replayed graph, per kernel and per launch grid (tools/prof_steady.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4final
mkdir -p $out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench.err || exit 1
tail -c 600 $out/bench_line.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/roofline -- python3 bench.py --roofline-only > $out/roofline_leg.json 2> $out/roofline_leg.err || exit 1
cp $out/roofline/*/*kernel_stats.csv $out/roofline_leg_kernel_stats.csv
rm -rf $out/roofline
bash tools/pmc_traffic.sh fwd > $out/traffic_fwd.txt 2>&1
bash tools/pmc_traffic.sh bwd > $out/traffic_bwd.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/replay -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > $out/replay_bench.json 2> $out/replay.err || exit 1
python3 tools/prof_steady.py $out/replay 1 120 "chunk_|scan_fwd_stream|gemm_|morph_|nf_|conv1d_|conv3x3|conv_s2|mamba_small|tri_" > $out/graph_replay_summary.txt 2>&1
rm -rf $out/replay
head -3 $out/graph_replay_summary.txt; tail -2 $out/traffic_fwd.txt; tail -2 $out/traffic_bwd.txt
