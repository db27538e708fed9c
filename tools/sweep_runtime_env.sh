#!/bin/bash
# The replayed training-step graph under a few HIP-runtime settings (gpurun): ms per step of bench.py --steps 20.
cd "$GRAFT_REPO_ROOT"
run() { env "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" 2>&1 | tail -1; }
for s in "X=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=1000" "DEBUG_HIP_GRAPH_BATCH_SIZE=10" "GPU_MAX_HW_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "AMD_DIRECT_DISPATCH=0" "HSA_ENABLE_INTERRUPT=0" "DEBUG_HIP_KERNARG_COPY_OPT=0"; do
  echo "$s -> $(run $s)"
done
