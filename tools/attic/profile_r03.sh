#!/bin/bash
# Round-3 step profile (run on the GPU box through gpurun): the default bench line, then a kernel trace of the
# replayed graph (8 steady steps) summarised per kernel by tools/prof_steady.py.  Outputs under gpurun_out/r3/<tag>/.
tag=${1:-step}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3/$tag
mkdir -p $out
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $out/bench_line.json 2> $out/bench.err || exit 1
cat $out/bench_line.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'img/s', d['value'])"
rocprofv3 --kernel-trace --output-format csv -d $out/replay -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > $out/replay_bench.json 2> $out/replay.err || exit 1
python3 tools/prof_steady.py $out/replay 1 90 "${2:-chunk_apply_bwd_w8|scan_fwd_stream|chunk_reduce8|gemm_nt_wide|gemm_tokens_mfma|morph_|nf_|conv1d_|conv3x3s_}" > $out/graph_replay_summary.txt 2>&1
rm -rf $out/replay
head -4 $out/graph_replay_summary.txt; tail -1 $out/graph_replay_summary.txt
