#!/bin/bash
# BASELINE configs 3 and 5 through bench.py, each once timed and once under rocprofv3 --kernel-trace (eager) for the
# per-kernel split.  Output under gpurun_out/<tag> (default r3cfg).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r3cfg}
mkdir -p $out
python3 bench.py --dtype bf16 --batch 16 --no-cpu-baseline --no-roofline > $out/c3.json 2> $out/c3.err
python3 bench.py --size 1024 --batch 2 --dtype bf16 --d-state 64 --no-cpu-baseline --no-roofline --steps 3 > $out/c5.json 2> $out/c5.err
python3 bench.py --infer --no-cpu-baseline --no-roofline > $out/c2.json 2> $out/c2.err
rocprofv3 --kernel-trace --output-format csv -d $out/c3t -- python3 bench.py --dtype bf16 --batch 16 --no-graph --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $out/c3t.err
python3 tools/prof_steady.py $out/c3t 1 40 > $out/c3_steady.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/c5t -- python3 bench.py --size 1024 --batch 2 --dtype bf16 --d-state 64 --no-graph --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $out/c5t.err
python3 tools/prof_steady.py $out/c5t 1 40 > $out/c5_steady.txt 2>&1
cat $out/c2.json $out/c3.json $out/c5.json | cut -c1-260
