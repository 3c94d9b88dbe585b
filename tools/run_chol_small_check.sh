#!/bin/bash
# one-launch solve of small systems against a host Cholesky, then the same matrices through the per-panel path
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out build
[ -x build/chol_small_check ] && [ build/chol_small_check -nt tools/chol_small_check.hip ] || /opt/rocm/bin/hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-function \
  -o build/chol_small_check tools/chol_small_check.hip -L/opt/rocm/lib -lrocsolver -lrocblas
build/chol_small_check small 2>&1 | tee gpurun_out/chol_small_check.txt
build/chol_small_check panels 2>&1 | tee -a gpurun_out/chol_small_check.txt
