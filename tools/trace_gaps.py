"""Idle time on the GPU inside one prove, from a rocprofv3 --kernel-trace results .db: the union of all kernel intervals against the
wall time between two PoW grinds (one per prove), and the largest gaps with the kernels on either side.
usage: python tools/trace_gaps.py results.db [n_gaps [which]]   which: 1 = the last prove in the trace (bench.py: the instrumented one, it
synchronises at every stage boundary), 2 = the one before it (a timed step), ..."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
n_gaps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(cur.execute("select start, end, name from kernels order by start"))
grinds = [i for i, r in enumerate(rows) if "grind_kernel" in r[2]]
if len(grinds) < 2:
    sys.exit("need two proves in the trace")
which = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if len(grinds) < which + 1:
    sys.exit("not enough proves in the trace")
a, b = grinds[-which - 1] + 1, grinds[-which] + 1
prove = rows[a:b]
t0, t1 = prove[0][0], max(r[1] for r in prove)
busy, ce = 0, prove[0][0]
gaps = []
prev_name = "(start)"
for s, e, name in prove:
    if s > ce:
        gaps.append((s - ce, prev_name, name, ce - t0))
        busy += 0
    if e > ce:
        busy += e - max(s, ce)
        prev_name = name if e >= ce else prev_name
        ce = e
print("one prove: %d launches, wall %.3f ms, GPU busy (union) %.3f ms, idle %.3f ms in %d gaps" % (len(prove), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps)))
hist = {}
for g, p, n, at in gaps:
    k = "<2us" if g < 2000 else "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"
    hist.setdefault(k, [0, 0]); hist[k][0] += 1; hist[k][1] += g
print("gap histogram:", {k: (v[0], round(v[1] / 1e6, 3)) for k, v in hist.items()})
print("largest gaps (us, at ms, after kernel -> before kernel):")
for g, p, n, at in sorted(gaps, reverse=True)[:n_gaps]:
    print("  %8.1f  @%7.3f  %s -> %s" % (g / 1e3, at / 1e6, p[:48], n[:48]))

# optional: list the launches of a window [from_ms, to_ms) of that prove, consecutive launches of one kernel merged
if len(sys.argv) > 5:
    lo, hi = float(sys.argv[4]) * 1e6 + t0, float(sys.argv[5]) * 1e6 + t0
    print("launches in [%s, %s) ms:" % (sys.argv[4], sys.argv[5]))
    run = None
    for s, e, name in prove:
        if s < lo or s >= hi:
            continue
        if run and run[0] == name:
            run[1] += 1; run[2] += e - s; run[4] = e
        else:
            if run: print("  @%7.3f  %4d x %-60s %9.1f us busy, %9.1f us span" % ((run[3] - t0) / 1e6, run[1], run[0][:60], run[2] / 1e3, (run[4] - run[3]) / 1e3))
            run = [name, 1, e - s, s, e]
    if run: print("  @%7.3f  %4d x %-60s %9.1f us busy, %9.1f us span" % ((run[3] - t0) / 1e6, run[1], run[0][:60], run[2] / 1e3, (run[4] - run[3]) / 1e3))
