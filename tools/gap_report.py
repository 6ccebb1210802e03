"""GPU idle time of one prove from a rocprofv3 --kernel-trace database: the gaps between consecutive kernels (any stream), grouped by
the kernel that ends before the gap and the one that starts after it.  python tools/gap_report.py results.db [n_proves=5]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
n_proves = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = list(cur.execute("select start, end, name from kernels order by start"))
grinds = [i for i, r in enumerate(rows) if "grind_kernel" in r[2]]
# the last prove: from just after the previous grind's decommit gather to this prove's gather
lo = grinds[-2] + 2 if len(grinds) >= 2 else 0
last = rows[lo:]
span = last[-1][1] - last[0][0]
cover, cs, ce = 0, last[0][0], last[0][1]
gaps = {}
prev_end, prev_name = last[0][1], last[0][2]
for s, e, name in last[1:]:
    if s > ce:
        g = s - ce
        key = (prev_name.split("(")[0][-40:], name.split("(")[0][-40:])
        a = gaps.setdefault(key, [0, 0]); a[0] += g; a[1] += 1
        cover += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
    if e >= prev_end: prev_end, prev_name = e, name
cover += ce - cs
print("last prove: %d kernels, span %.3f ms, GPU busy (union) %.3f ms, idle %.3f ms" % (len(last), span / 1e6, cover / 1e6, (span - cover) / 1e6))
for (a, b), (t, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%8.1f us  x%-4d  %s  ->  %s" % (t / 1e3, n, a, b))
