"""Turns a rocprofv3 (--kernel-trace --stats) results .db into a small text summary for profiles/."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
print("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)", file=out)
print("# source:", sys.argv[1], file=out)
if len(sys.argv) > 3: print("# command:", sys.argv[3], file=out)
print("%-70s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"), file=out)
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-70s %8d %14.1f %12.3f %7.2f" % (name[:70], calls, total / 1.0, avg, pct), file=out)
print("\n# per launch shape (grid threads, block, dynamic LDS bytes): calls, avg_us", file=out)
for r in cur.execute("select name, grid_x, workgroup_x, lds_size, count(*), avg(duration)/1000.0 from kernels group by name, grid_x, workgroup_x, lds_size having avg(duration) > 20000 order by 1, 6 desc"):
    print("%-50s grid=%-10d block=%-5d lds=%-6d calls=%-5d avg_us=%.1f" % (r[0][:50], r[1], r[2], r[3], r[4], r[5]), file=out)

# The LDE launches (fft13_kernel, lde_mid_kernel) of a column batch run on two streams and overlap: the sum of their durations double-counts the time they
# occupy.  bench.py's roofline.kernel_ms is the wall time of the LDE work (HIP events around the stream fork/join); the matching
# figure from this trace is the UNION of the fft13 kernel intervals.
rows = list(cur.execute("select start, end from kernels where name like '%fft13_kernel%' or name like '%lde_mid_kernel%' order by start"))
if rows:
    total = sum(b - a for a, b in rows)
    union, cs, ce = 0, rows[0][0], rows[0][1]
    for a, b in rows[1:]:
        if a > ce: union += ce - cs; cs, ce = a, b
        else: ce = max(ce, b)
    union += ce - cs
    proves = max(1, cur.execute("select count(*) from kernels where name like '%grind_kernel%'").fetchone()[0])   # one PoW grind per prove
    print("\n# LDE kernels (fft13_kernel passes + lde_mid_kernel): %d launches, sum of durations %.1f us, union of their intervals %.1f us (%d proves -> %.2f ms of LDE wall time per prove; "
          "bench.py roofline.kernel_ms measures this with HIP events)" % (len(rows), total / 1e3, union / 1e3, proves, union / 1e6 / proves), file=out)
