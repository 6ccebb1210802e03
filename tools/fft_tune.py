"""Micro-benchmark of the fused LDE (iFFT + FFT) and Merkle commit for tuning; prints one JSON line per option set.
usage: fft_tune.py LOG NCOLS REPS [name=value,name=value ...]   each further argument is one set of nx_ctx_set_option settings
(e.g. fft.batch_cols=4,fft.streams=1  fft.batch_cols=2,fft.streams=2); without any, the context's defaults."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nexus_zkvm_amd as nz

log = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sets = sys.argv[4:] or [""]
be = nz.HipBackend(0)
tw = be.precompute_twiddles(log)
if os.environ.get("FFT_TUNE_BYTE_LIMBS") == "1":   # config #2's byte-limb variant: values in [0, 256) like the reference's limb columns (prover/src/trace/utils.rs:57-62)
    cols = be.columns(ncols, log)
    _rng = np.random.default_rng(3)
    for _c0 in range(0, ncols, 32):
        _n = min(32, ncols - _c0)
        _blk = _rng.integers(0, 256, (_n, 1 << log), dtype=np.uint32)
        be._chk(be.L.nx_upload(be.ctx, __import__("ctypes").c_void_p(cols.ptr.value + _c0 * (4 << log)), _blk.ctypes.data_as(__import__("ctypes").c_void_p), __import__("ctypes").c_size_t(_blk.size)))
else:
    cols = be.synth_fill_tree([(log, 2, ncols, 0)], 1, seed=3)[0]
out = be.columns(ncols, log + 1)
import ctypes as _C
if os.environ.get("FFT_TUNE_ZEROS") == "1":      # DVFS probe: all-zero data toggles far fewer bits (MI355X_MICROARCH.md: the chip clocks to its power budget)
    be._chk(be.L.nx_memset_zero(be.ctx, cols.ptr, _C.c_size_t(ncols << log)))
be.sync()
defaults = {k: be.get_option(k) for k in ("fft.batch_cols", "fft.streams")}
rounds = int(os.environ.get("FFT_TUNE_ROUNDS", "1"))     # > 1: the option sets are cycled A/B/C/A/B/C ... and min / median are reported per set
samples = {st: [] for st in sets}
for st in (sets * rounds if rounds > 1 else []):
    for k, v in defaults.items():
        be.set_option(k, v)
    for k, v in dict(kv.split("=") for kv in st.split(",") if kv).items():
        be.set_option(k, int(v))
    for r in range(reps):
        be.sync(); t0 = time.perf_counter()
        be._chk(be.L.nx_lde_batch(be.ctx, tw.h, cols.col_ptrs(), ncols, log, 1, out.col_ptrs()))
        be.sync(); samples[st].append(time.perf_counter() - t0)
if rounds > 1:
    for st in sets:
        v = sorted(samples[st][1:])
        print(json.dumps({"opts": st, "log": log, "ncols": ncols, "rounds": rounds, "min_ms": round(v[0] * 1e3, 3), "median_ms": round(v[len(v) // 2] * 1e3, 3),
                          "median_frac_of_8TBs": round(ncols * 16 * (1 << log) / v[len(v) // 2] / 8e12, 4)}), flush=True)
    sets = []
for st in sets:
    for k, v in defaults.items():
        be.set_option(k, v)
    opts = dict(kv.split("=") for kv in st.split(",") if kv)
    for k, v in opts.items():
        be.set_option(k, int(v))
    best = 1e9
    for r in range(reps + 1):
        be.sync(); t0 = time.perf_counter()
        be._chk(be.L.nx_lde_batch(be.ctx, tw.h, cols.col_ptrs(), ncols, log, 1, out.col_ptrs()))
        be.sync(); dt = time.perf_counter() - t0
        if r: best = min(best, dt)
    alg = ncols * 16 * (1 << log)
    if os.environ.get("FFT_TUNE_ZEROS") == "1":
        be._chk(be.L.nx_memset_zero(be.ctx, cols.ptr, _C.c_size_t(ncols << log)))   # the LDE overwrote the columns with coefficients (zero stays zero)
    print(json.dumps({"opts": st, "zeros": os.environ.get("FFT_TUNE_ZEROS") == "1", "log": log, "ncols": ncols, "lde_ms": round(best * 1e3, 3), "lde_alg_GBs": round(alg / best / 1e9, 1),
                      "frac_of_8TBs": round(alg / best / 8e12, 4)}), flush=True)
if os.environ.get("FFT_TUNE_MERKLE", "1") != "0":
    bm = 1e9
    for r in range(reps):
        be.sync(); t0 = time.perf_counter()
        t = be.merkle_commit([out]); be.sync(); dt = time.perf_counter() - t0
        bm = min(bm, dt)
    malg = ncols * 4 * (2 << log) + 128 * (2 << log)
    print(json.dumps({"log": log, "ncols": ncols, "merkle_ms": round(bm * 1e3, 3), "merkle_alg_GBs": round(malg / bm / 1e9, 1)}))
