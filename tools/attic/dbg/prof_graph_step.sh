#!/bin/bash
# kernel trace of the graph-replayed training step (the default bench mode): what a replay launches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/r2b/graphstep}
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/run -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $out/bench.json 2> $out/err.txt
python3 tools/prof_steady.py $out/run 1 70 > $out/summary.txt 2>&1
head -75 $out/summary.txt | cut -c1-200
