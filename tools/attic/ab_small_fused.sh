cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3
for v in 1 0 1 0; do
  MMUNET_SMALL_FUSED=$v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused=$v ms_per_step', d['ms_per_step'])"
done
