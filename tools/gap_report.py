#!/usr/bin/env python3
"""Development aid: per-LM-iteration GPU busy fraction and idle gaps from a rocprofv3 rocpd database."""
import collections
import glob
import re
import sqlite3
import sys


def main(path):
    db = glob.glob(path + '/**/*.db', recursive=True)[0] if not path.endswith('.db') else path
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = list(cur.execute(f"select d.start,d.end,s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    names = [r[2] for r in rows]
    lp = [i for i, n in enumerate(names) if 'k_linearize_all' in n]
    a, b = lp[60], lp[min(360, len(lp) - 1)]
    nit = min(360, len(lp) - 1) - 60
    t0, t1 = rows[a][0], rows[b][0]
    busy, cur_end, gaps = 0, t0, []
    for s, e, n in rows[a:b]:
        if s > cur_end:
            gaps.append((s - cur_end, n))
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
    print("per iteration %.1f us, busy %.3f, idle per iteration %.1f us" % ((t1 - t0) / 1e3 / nit, busy / (t1 - t0), sum(d for d, _ in gaps) / 1e3 / nit))
    g = collections.defaultdict(lambda: [0, 0])
    for d, n in gaps:
        k = re.sub(r'^_ZN3ppo\d*', '', n)[:40]
        g[k][0] += d
        g[k][1] += 1
    for k, v in sorted(g.items(), key=lambda kv: -kv[1][0])[:10]:
        print("  idle before %-42s %8.1f us/iter  n/iter=%.2f  avg %.2f us" % (k, v[0] / 1e3 / nit, v[1] / nit, v[0] / 1e3 / v[1]))
    # one iteration's timeline
    i0 = lp[100]
    i1 = lp[101]
    base = rows[i0][0]
    for s, e, n in rows[i0 - 3:i1]:
        print("    %9.1f %8.1f  %s" % ((s - base) / 1e3, (e - s) / 1e3, re.sub(r'^_ZN3ppo\d*', '', n)[:50]))


if __name__ == "__main__":
    main(sys.argv[1])
