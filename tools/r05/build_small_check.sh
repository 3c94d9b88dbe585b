#!/bin/bash
# builds tools/chol_small_check.hip into build/ and prints the register / LDS use of chol_small
cd "$(dirname "$0")/../.." && mkdir -p build /tmp/chk
/opt/rocm/bin/hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -o build/chol_small_check tools/chol_small_check.hip -L/opt/rocm/lib -lrocsolver -lrocblas -save-temps=obj || exit 1
python3 - <<'PY'
import re,glob
for f in glob.glob('build/*gfx950*.s'):
    s=open(f).read()
    for m in re.finditer(r'  - \.agpr_count:.*?\.wavefront_size: *\d+', s, re.S):
        blk=m.group(0)
        name=re.search(r'\.name: *(\S+)',blk).group(1)
        if 'small' not in name: continue
        g=lambda k: re.search(r'\.%s: *(\d+)'%k,blk).group(1)
        print(name[:60], 'vgpr',g('vgpr_count'),'agpr',g('agpr_count'),'sgpr',g('sgpr_count'),'lds',g('group_segment_fixed_size'),'scratch',g('private_segment_fixed_size'),'spill',g('vgpr_spill_count'))
PY
cp build/chol_small_check-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/chk/small.s; rm -f build/chol_small_check-*
/opt/rocm/bin/hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -DNO_STAMPS -munsafe-fp-atomics -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -o build/chol_small_check_nostamps tools/chol_small_check.hip -L/opt/rocm/lib -lrocsolver -lrocblas
