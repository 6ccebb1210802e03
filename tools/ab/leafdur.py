import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select start, end, name, grid_x from kernels order by start"))
grinds = [i for i, r in enumerate(rows) if "grind_kernel" in r[2]]
a, b = grinds[-3] + 1, grinds[-2] + 1
t0 = rows[a][0]
for s, e, n, g in rows[a:b]:
    if "merkle_leaf_chain" in n or ("merkle_layer" in n and g >= 4194304):
        # what else runs concurrently?
        ov = [r[2][:30] for r in rows[a:b] if r[0] < e and r[1] > s and r[2] != n]
        print("@%7.3f ms  %-34s grid=%-9d %8.1f us  concurrent: %d %s" % ((s - t0) / 1e6, n[9:43], g, (e - s) / 1e3, len(ov), sorted(set(ov))[:2]))
