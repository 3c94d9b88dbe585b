#!/bin/bash
# Review item 3 (two-CU chol_small for 97..192 unknowns), bounded before building: the one-launch solve with column k applied only to the tile
# columns k+1..k+c of the owners (everything further right "done by a helper compute unit for free"; results are wrong, only the clock is read),
# for c = 1 (the bare chain), 2 and 3 columns of slack, next to the real kernel and to the panel path.  Output: gpurun_out/r06_two_cu_bound.txt
cd "$(dirname "$0")/../.." || exit 1
mkdir -p build gpurun_out
FLAGS="-w -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -DNO_STAMPS"
for c in 0 1 2 3; do
  def=""; [ $c -gt 0 ] && def="-DSM_TWO_CU_BOUND=$c"
  [ -x build/two_cu_bound_$c ] || /opt/rocm/bin/hipcc $FLAGS $def -o build/two_cu_bound_$c tools/chol_small_check.hip -L/opt/rocm/lib -lrocsolver -lrocblas &
done
wait
for c in 0 1 2 3; do [ -x build/two_cu_bound_$c ] || exit 1; done
out=gpurun_out/r06_two_cu_bound.txt
: > $out
for c in 0 1 2 3; do
  echo "== one launch, owners apply column k to tile columns k+1..k+$c only (0 = the real kernel)" >> $out
  timeout 120 build/two_cu_bound_$c small 2>&1 | grep -E "^n +(96|1[0-9][0-9]) " | grep -v "leading minor" >> $out
done
echo "== panel path" >> $out
timeout 120 build/two_cu_bound_0 panels 2>&1 | grep -E "^n +(96|1[0-9][0-9]) " | grep -v "leading minor" >> $out
cat $out
