#!/bin/bash
# development aid: per-kernel durations of the native dense solve alone (tools/chol_check.hip, per-panel arm) with and without the
# inverse blocks of the backward substitution
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -DPPO_BA_TEST_HOOKS -munsafe-fp-atomics -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -o /tmp/chol_check chol_check.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/choltrace
  PPO_BA_CHOL_NO_INV=$v rocprofv3 --kernel-trace --stats -d /tmp/choltrace -o p -- /tmp/chol_check panels > /dev/null 2>&1
  echo "PPO_BA_CHOL_NO_INV=$v"
  python3 - <<'PY'
import glob, sqlite3
db = glob.glob('/tmp/choltrace/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select d.start,d.end,s.kernel_name,d.grid_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
# the last solve at ld = 1216: find the last run of 19 chol_step + 1 back with grid pattern
idx = [i for i, r in enumerate(rows) if 'chol_back_all' in r[2] and r[3] in (19 * 8 * 256, 19 * 8 * 320)]
i = idx[-1]
seq = rows[i - 19:i + 1]
print(' '.join('%.1f' % ((e - s) / 1e3) for s, e, n, g in seq), '| total %.1f us' % ((seq[-1][1] - seq[0][0]) / 1e3))
PY
done
