#!/bin/bash
# round 6: builds tools/potf2_anatomy.hip into build/ (cross-compiles without a GPU) and, on a GPU box, runs it -> gpurun_out/r06_potf2_anatomy.txt
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $ROOT/build $ROOT/gpurun_out
if [ ! -x $ROOT/build/potf2_anatomy ] || [ $ROOT/tools/potf2_anatomy.hip -nt $ROOT/build/potf2_anatomy ]; then
  /opt/rocm/bin/hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -I $ROOT/include -o $ROOT/build/potf2_anatomy $ROOT/tools/potf2_anatomy.hip
fi
if [ "$1" != "build" ]; then $ROOT/build/potf2_anatomy | tee $ROOT/gpurun_out/r06_potf2_anatomy.txt; fi
