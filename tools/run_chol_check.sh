#!/bin/bash
# builds and runs tools/chol_check.hip on the GPU box (development aid): one-launch factorisation, then the per-panel one
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -DPPO_BA_TEST_HOOKS -munsafe-fp-atomics -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -o /tmp/chol_check chol_check.hip
timeout 120 /tmp/chol_check
timeout 120 /tmp/chol_check panels
