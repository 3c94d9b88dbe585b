cd tools
for flags in "" "-DPPO_CHOL_FWD_ONE_XCD"; do
  /opt/rocm/bin/hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -DPPO_BA_TEST_HOOKS $flags -munsafe-fp-atomics -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -o /tmp/chol_check chol_check.hip
  echo "== one launch, flags: $flags"
  timeout 120 /tmp/chol_check 2>&1 | grep -E "^n " 
done
