#!/bin/bash
# builds and runs tools/r06/boundary_vs_barrier.hip (review item 4: launch boundary against grid barrier) -> gpurun_out/r06_boundary_vs_barrier.txt
cd "$(dirname "$0")/../.." || exit 1
mkdir -p build gpurun_out
/opt/rocm/bin/hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -o build/boundary_vs_barrier tools/r06/boundary_vs_barrier.hip || exit 1
timeout 120 build/boundary_vs_barrier 2>&1 | tee gpurun_out/r06_boundary_vs_barrier.txt
