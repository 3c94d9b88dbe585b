#!/bin/bash
# builds and runs tools/chol_probe.hip on the GPU box (development aid)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -o /tmp/chol_probe chol_probe.hip
/tmp/chol_probe
