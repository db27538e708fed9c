#!/bin/bash
# A/B over a compile-time switch of ONE translation unit (run on the GPU box): builds csrc/<unit>.hip with -D<macro>=<v>
# for each value, relinks, runs the command; leaves the library as build.sh makes it.
# usage: tools/ab_obj.sh <unit> <macro> "<command>" v1 v2 ...
set -e
unit=$1; macro=$2; cmd=$3; shift 3
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/mm-unet_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -Wno-unused-function"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -D$macro=$v -c $unit.hip -o $unit.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmmunet_hip.so *.o
  echo "== $macro=$v"
  (cd "$root" && eval "$cmd")
done
rm -f $unit.o.hash
bash build.sh > /dev/null
