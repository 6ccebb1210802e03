"""What the host does while the GPU idles inside one prove: rocprofv3 --hip-trace --kernel-trace results .db -> for every idle gap of
at least MIN_US (union of kernel intervals, the prove before the last PoW grind = a timed step of bench.py) the HIP API calls that
overlap it, with their durations, and the host time inside the gap that no HIP call covers (the library's own C++).
usage: python tools/gap_api.py results.db [min_us]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows = list(cur.execute("select start, end, name from kernels order by start"))
grinds = [i for i, r in enumerate(rows) if "grind_kernel" in r[2]]
a, b = grinds[-3] + 1, grinds[-2] + 1
prove = rows[a:b]
t0, t1 = prove[0][0], max(r[1] for r in prove)
api = list(cur.execute("select name, start, end from regions where end >= ? and start <= ? order by start", (t0 - 500000, t1 + 500000)))
gaps, ce, prev = [], prove[0][0], "(start)"
# the hand-over from the previous prove: its grind's end to this prove's first kernel
gaps.append((rows[a - 1][1], prove[0][0], rows[a - 1][2], prove[0][2]))
for s, e, name in prove:
    if s > ce: gaps.append((ce, s, prev, name))
    if e > ce: ce, prev = e, name
idle = sum(g[1] - g[0] for g in gaps)
print("one prove: %.3f ms wall, %d launches, idle %.3f ms in %d gaps (incl. the hand-over from the previous prove)" % ((t1 - t0) / 1e6, len(prove), idle / 1e6, len(gaps)))
for g0, g1, pa, pb in sorted(gaps, key=lambda g: g[0] - g[1]):
    if (g1 - g0) / 1e3 < min_us: continue
    print("\ngap %.1f us at %.3f ms   %s -> %s" % ((g1 - g0) / 1e3, (g0 - t0) / 1e6, pa[:60], pb[:60]))
    covered, last = 0, g0
    inside = [(n, max(s, g0), min(e, g1), e - s) for n, s, e in api if e > g0 and s < g1]
    for n, s, e, full in inside:
        if e > last: covered += e - max(s, last); last = max(last, e)
    agg = {}
    for n, s, e, full in inside:
        k = agg.setdefault(n, [0, 0.0]); k[0] += 1; k[1] += (e - s) / 1e3
    for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]: print("    %7.1f us %4d x %s" % (us, c, n))
    print("    %7.1f us outside any HIP call" % ((g1 - g0 - covered) / 1e3))
