#!/bin/bash
# Same-box A/B of the working tree against a git revision (boxes differ by +-0.4 ms, one box repeats to +-0.03):
#   here:        tools/ab_bench.sh prepare <rev>     exports <rev> into ab_prev/ and builds its library (CPU cross-compile)
#   on the box:  tools/ab_bench.sh run [n]           alternates `bench.py --gpus 1 --steps 20 --warmup 5` of both trees n times
# ab_prev/ is untracked scratch (listed in .gitignore); delete it afterwards -- it travels with every gpurun push.
set -e
cd "$(dirname "$0")/.."
if [ "$1" = prepare ]; then
    rm -rf ab_prev && mkdir ab_prev
    git archive "$2" | tar -x -C ab_prev
    bash ab_prev/mm-unet_amd/csrc/build.sh | tail -1
    [ -d ab_prev/mm_unet_amd ] || cp -a mm_unet_amd ab_prev/ 2>/dev/null || true
elif [ "$1" = run ]; then
    for i in $(seq 1 "${2:-2}"); do
        for t in . ab_prev; do
            ( cd $t && python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 |
              python3 -c "import json,sys; print('$t', json.loads(sys.stdin.read())['ms_per_step'])" )
        done
    done
fi
