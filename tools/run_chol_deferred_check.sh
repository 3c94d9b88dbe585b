#!/bin/bash
# builds and runs tools/chol_deferred_check.hip on the GPU box (development aid)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -o /tmp/chol_deferred_check chol_deferred_check.hip
timeout 120 /tmp/chol_deferred_check
