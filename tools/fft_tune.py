"""Micro-benchmark of the fused LDE (iFFT + FFT) and Merkle commit for tuning; prints one JSON line.
Env knobs (read once per process by fft.hip): NX_FFT_SMAX, NX_FFT_B, NX_FFT_THREADS, NX_FFT_BATCH."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nexus_zkvm_amd as nz

log = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
be = nz.HipBackend(0)
tw = be.precompute_twiddles(log)
cols = be.synth_fill_tree([(log, 2, ncols, 0)], 1, seed=3)[0]
out = be.columns(ncols, log + 1)
be.sync()
best = 1e9
for r in range(reps + 1):
    be.sync(); t0 = time.perf_counter()
    be._chk(be.L.nx_lde_batch(be.ctx, tw.h, cols.col_ptrs(), ncols, log, 1, out.col_ptrs()))
    be.sync(); dt = time.perf_counter() - t0
    if r: best = min(best, dt)
alg = ncols * 16 * (1 << log)
bm = 1e9
for r in range(reps):
    be.sync(); t0 = time.perf_counter()
    t = be.merkle_commit([out]); be.sync(); dt = time.perf_counter() - t0
    bm = min(bm, dt)
malg = ncols * 4 * (2 << log) + 128 * (2 << log)
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("NX_")}, "log": log, "ncols": ncols, "lde_ms": best * 1e3,
                  "lde_alg_GBs": alg / best / 1e9, "lde_frac_of_8TBs": alg / best / 8e12, "lde_frac_of_measured_copy_6.29TBs": alg / best / 6.29e12, "merkle_ms": bm * 1e3, "merkle_alg_GBs": malg / bm / 1e9}))
