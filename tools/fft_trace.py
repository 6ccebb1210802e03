"""Per-block phase timeline of the fft13 kernels (needs the -DNX_FFT_TRACE build: tools/build_trace_lib.sh, NX_LIB=...).
Prints, per kernel kind, the mean time a block spends in staging (global load -> LDS, incl. the fused edge layer), the LDS rounds,
issuing the stores, and waiting for the store acknowledgements (100 MHz wall clock)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nexus_zkvm_amd as nz
log = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 64
be = nz.HipBackend(0)
tw = be.precompute_twiddles(log)
cols = be.synth_fill_tree([(log, 2, ncols, 0)], 1, seed=3)[0]
out = be.columns(ncols, log + 1)
buf = (C.c_ulonglong * 32)()
for r in range(2):
    be.sync(); be.L.nx_fft13_trace_read(buf, 1)
    be._chk(be.L.nx_lde_batch(be.ctx, tw.h, cols.col_ptrs(), ncols, log, 1, out.col_ptrs()))
    be.sync()
be.L.nx_fft13_trace_read(buf, 0)
a = np.array(list(buf), dtype=np.float64).reshape(4, 8)
names = {0: "fwd pass (layers>=13)", 1: "fwd FIRST (layers 0..12)", 2: "inv pass (layers>=13)", 3: "inv FIRST (layers 0..12)"}
for k in range(4):
    n = a[k, 7]
    if n:
        us = a[k, :4] / n / 100.0
        print(json.dumps({"kernel": names[k], "blocks": int(n), "stage_us": round(us[0], 2), "rounds_us": round(us[1], 2), "store_issue_us": round(us[2], 2),
                          "store_ack_us": round(us[3], 2), "total_us": round(us.sum(), 2)}))
