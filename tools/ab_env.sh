#!/bin/bash
# Same-box comparison of one environment variable's values on the training step:
#   tools/ab_env.sh VAR v1 v2 ...   -> ms per step of `bench.py --gpus 1 --steps 20 --warmup 5` for each value, twice, interleaved
var=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    ms=$(env $var=$v python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 |
         python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$var=$v $ms"
  done
done
