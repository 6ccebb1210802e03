"""Is the LDE power-limited?  Runs nx_lde_batch in a loop for a few seconds on random and on all-zero columns while sampling
rocm-smi (socket power, sclk) from a side thread; prints one JSON line per data kind.  MI355X_MICROARCH.md: "the chip clocks to its
power budget ... zero-filled inputs ran +19 %"."""
import ctypes as C, json, os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nexus_zkvm_amd as nz

log, ncols, secs = 22, 128, float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
be = nz.HipBackend(0)
tw = be.precompute_twiddles(log)
cols = be.synth_fill_tree([(log, 2, ncols, 0)], 1, seed=3)[0]
out = be.columns(ncols, log + 1)
be.sync()


def sample(stop, acc):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Power \(W\):\s*([0-9.]+)", t)
            c = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", t)
            if p and c:
                acc.append((float(p.group(1)), int(c.group(1))))
        except Exception:
            pass
        time.sleep(0.2)


what = sys.argv[2] if len(sys.argv) > 2 else "lde"      # lde | merkle (Blake2s commit of the LDE columns)
for kind in ("random", "zeros", "random"):
    if kind == "zeros":
        be._chk(be.L.nx_memset_zero(be.ctx, cols.ptr, C.c_size_t(ncols << log)))
    else:
        cols = be.synth_fill_tree([(log, 2, ncols, 0)], 1, seed=3)[0]
    be.sync()
    stop, acc = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, acc)); th.start()
    n, t0 = 0, time.perf_counter()
    if what == "merkle":
        be._chk(be.L.nx_lde_batch(be.ctx, tw.h, cols.col_ptrs(), ncols, log, 1, out.col_ptrs())); be.sync()   # zeros stay zeros
    while time.perf_counter() - t0 < secs:
        if what == "merkle":
            t = be.merkle_commit([out]); be.sync(); n += 1
            continue
        be._chk(be.L.nx_lde_batch(be.ctx, tw.h, cols.col_ptrs(), ncols, log, 1, out.col_ptrs()))   # coefficients of coefficients: still "random" words
        be.sync(); n += 1
    el = time.perf_counter() - t0
    stop.set(); th.join()
    acc = acc[2:] or acc
    print(json.dumps({"what": what, "data": kind, "ms": round(1e3 * el / n, 3), "alg_GBs": round(ncols * 16 * (1 << log) * n / el / 1e9, 1), "samples": len(acc),
                      "avg_power_W": round(sum(a for a, _ in acc) / max(1, len(acc)), 1), "avg_sclk_MHz": round(sum(b for _, b in acc) / max(1, len(acc)))}), flush=True)
