#!/bin/bash
# timing experiments on conv3x3_mfma (run on the GPU box): builds the one object with -DMMU_CONV3_EXP=n, relinks, times
# bench.py's conv leg.  Leaves the library as build.sh makes it.
set -e
cd "$(dirname "$0")/../mm-unet_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -Wno-unused-function"
for e in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DMMU_CONV3_EXP=$e -c conv3x3_mfma.hip -o conv3x3_mfma.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmmunet_hip.so *.o
  (cd ../.. && python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('EXP=$e', [(s['input'], s['ms_per_launch']) for s in d['roofline_conv']['shapes']])")
done
rm -f conv3x3_mfma.o.hash
bash build.sh > /dev/null
