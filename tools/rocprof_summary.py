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
