#!/bin/bash
# Round-2 profile set (run on the GPU box through gpurun; summaries land under gpurun_out/r2prof/, the ones kept
# for the record are copied to profiles/ by hand):
#   1. rocprofv3 --kernel-trace --stats of `bench.py --roofline-only`        (kernel averages behind the roofline legs)
#   2. SQ counters of the scan forward / backward kernels                     (tools/pmc_scan_fwd.sh)
#   3. HBM traffic of one scan forward / backward call                        (tools/pmc_traffic.sh)
#   4. eager kernel trace of three training steps + steady-state summary      (tools/prof_steady.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2prof
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/roofline -- python3 bench.py --roofline-only > $out/roofline_leg.json 2> $out/roofline_leg.err
cp $out/roofline/*/*kernel_stats.csv $out/roofline_leg_kernel_stats.csv
bash tools/pmc_scan_fwd.sh r2fwd > $out/pmc_fwd.txt 2>&1
PROF_SCRIPT=tools/prof_scan_bwd.py bash tools/pmc_scan_fwd.sh r2bwd > $out/pmc_bwd.txt 2>&1
bash tools/pmc_traffic.sh fwd > $out/traffic_fwd.txt 2>&1
bash tools/pmc_traffic.sh bwd > $out/traffic_bwd.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/steady -- python3 bench.py --no-graph --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $out/steady_bench.json 2> $out/steady.err
python3 tools/prof_steady.py $out/steady 1 60 > $out/steady_state_summary.txt 2>&1
tail -3 $out/pmc_fwd.txt; tail -3 $out/pmc_bwd.txt; head -5 $out/steady_state_summary.txt
