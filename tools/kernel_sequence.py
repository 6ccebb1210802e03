"""The kernels of the LAST prove of a rocprofv3 --kernel-trace run (between its last two grind kernels), in start order, with the GPU-idle
gap in front of each — where the host keeps the GPU waiting.  usage: kernel_sequence.py results.db out.txt
Trace tools/prove_loop.py for this, not bench.py / keccak_shaped.py: their LAST prove is the statistics prove, whose stages each end in a
synchronisation (round 6: such a trace showed 8.7 ms of idle time for a statement whose timed proves idle 3.7 ms)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
g = [i for i, r in enumerate(rows) if 'grind' in r[0]]
a, b = g[-2] + 1, g[-1] + 1
t0 = rows[a][1]
out = open(sys.argv[2], 'w')
prev_end, lines, idle = t0, [], 0.0
for r in rows[a:b]:
    gap = (r[1] - prev_end) / 1e3
    if gap > 0: idle += gap
    lines.append((gap, "%9.1f %8.1f gap %6.1f  %-60s grid=%d" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, r[0][:60], r[3])))
    prev_end = max(prev_end, r[2])
out.write("# one prove: %d kernels over %.1f us, GPU idle %.1f us; the 12 largest gaps (us, in front of):\n" % (b - a, (prev_end - t0) / 1e3, idle))
for gap, l in sorted(lines, key=lambda x: -x[0])[:12]:
    out.write("#   " + l + "\n")
for _, l in lines:
    out.write(l + "\n")
