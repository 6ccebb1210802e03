"""Where the host spends the time between kernels of one prove: rocprofv3 --hip-trace --kernel-trace results .db -> the HIP API calls
of the last prove by total duration, and the longest single calls (run on the GPU box; prints text)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
k = list(cur.execute("select name,start,end from kernels order by start"))
fills = [i for i, r in enumerate(k) if "synth_fill" in r[0]]
t0 = k[fills[-3]][1]; t1 = max(r[2] for r in k[fills[-3]:])
print("last prove: %.2f ms wall, %d kernels" % ((t1 - t0) / 1e6, len(k) - fills[-3]))
cols = [r[1] for r in cur.execute("pragma table_info(regions)")]
rows = list(cur.execute("select name,start,end from regions where start >= ? and end <= ? order by start", (t0 - 200000, t1 + 200000)))
by = collections.defaultdict(lambda: [0, 0])
for n, a, b in rows: by[n][0] += b - a; by[n][1] += 1
for n, (t, c) in sorted(by.items(), key=lambda x: -x[1][0])[:14]: print("%9.1f us %5d calls  %s" % (t / 1e3, c, n))
print("longest calls:")
for n, a, b in sorted(rows, key=lambda r: r[1] - r[2])[:25]: print("%8.1f us at %7.3f ms  %s" % ((b - a) / 1e3, (a - t0) / 1e6, n))
