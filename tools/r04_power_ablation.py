"""Where do the LDE's watts go?    nx_lde_batch of 128 columns x 2^22 rows in a loop, socket power and shader clock
sampled from rocm-smi, for:
  random        uniform M31 words (the bench's data)                                      — the baseline
  zeros         all-zero columns                                                          — no toggling anywhere (upper bound of what data can give)
  byte_limbs    values in [0, 256) (the reference's limb columns, trace/utils.rs:57-62)   — inputs small, coefficients full-width after the iFFT
  tw_one        random data, every twiddle = 1 (the doubled tables hold 2)                — the 32x32 multiplier sees one constant operand
  tw_one_zeros  zeros + twiddles 1
and, on random data, the column-batch shapes that decide whether the three launches of a batch hand over through the Infinity Cache:
  fft.batch_cols x fft.streams = 2x2 (default: 4 columns in flight, 4 x 48 MiB), 8x2, 32x1, 128x1 (no reuse on die).
The twiddle override pokes the library's tables through the handle (struct nx_twiddles: ctx, log_half, d_tw, d_itw, d_tw2, d_itw2) — results
are garbage by design, this is a timing / power probe only.   usage: python tools/r04_power_ablation.py [seconds per variant]"""
import ctypes as C, json, os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nexus_zkvm_amd as nz

log, ncols = 22, 128
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
be = nz.HipBackend(0)
out = be.columns(ncols, log + 1)


class TwStruct(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("log_half", C.c_uint32), ("d_tw", C.c_void_p), ("d_itw", C.c_void_p), ("d_tw2", C.c_void_p), ("d_itw2", C.c_void_p)]


def twiddles(ones):
    tw = be.precompute_twiddles(log)
    if ones:
        st = C.cast(tw.h, C.POINTER(TwStruct)).contents
        n = 1 << st.log_half
        one, two = np.ones(n, np.uint32), np.full(n, 2, np.uint32)
        for ptr, src in ((st.d_tw, one), (st.d_itw, one), (st.d_tw2, two), (st.d_itw2, two)):
            be._chk(be.L.nx_upload(be.ctx, C.c_void_p(ptr), src.ctypes.data_as(C.c_void_p), C.c_size_t(n)))
    return tw


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
        time.sleep(0.15)


def fill(kind):
    cols = be.synth_fill_tree([(log, 2, ncols, 0)], 1, seed=3)[0]
    if kind == "zeros":
        be._chk(be.L.nx_memset_zero(be.ctx, cols.ptr, C.c_size_t(ncols << log)))
    elif kind == "byte_limbs":
        row = np.random.default_rng(1).integers(0, 256, 1 << log, dtype=np.uint32)
        for c in range(ncols):
            be._chk(be.L.nx_upload(be.ctx, C.c_void_p(cols.ptr.value + c * (4 << log)), np.roll(row, c).ctypes.data_as(C.c_void_p), C.c_size_t(1 << log)))
    be.sync()
    return cols


def run(name, data, ones=False, opts=None):
    for k, v in (opts or {"fft.batch_cols": 2, "fft.streams": 2}).items():
        be.set_option(k, v)
    tw = twiddles(ones)
    cols = fill(data)
    src = be.clone_columns(cols)
    stop, acc = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, acc)); th.start()
    n, busy, t0 = 0, 0.0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        if data != "random" or ones:       # the transform overwrites the columns with coefficients: restore the probe's input (not timed)
            be._chk(be.L.nx_copy(be.ctx, cols.ptr, src.ptr, C.c_size_t(ncols << log))); be.sync()
        t1 = time.perf_counter()
        be._chk(be.L.nx_lde_batch(be.ctx, tw.h, cols.col_ptrs(), ncols, log, 1, out.col_ptrs()))
        be.sync(); busy += time.perf_counter() - t1; n += 1
    stop.set(); th.join()
    acc = acc[2:] or acc
    print(json.dumps({"variant": name, "data": data, "twiddles_one": ones, "opts": opts or "default (2 x 2)", "lde_ms": round(1e3 * busy / n, 3),
                      "alg_GBs": round(ncols * 16 * (1 << log) * n / busy / 1e9, 1), "duty": round(busy / (time.perf_counter() - t0), 2), "samples": len(acc),
                      "avg_power_W": round(sum(a for a, _ in acc) / max(1, len(acc)), 1), "avg_sclk_MHz": round(sum(b for _, b in acc) / max(1, len(acc)))}), flush=True)
    cols.free(); src.free()


run("random", "random")
run("zeros", "zeros")
run("byte_limbs", "byte_limbs")
run("tw_one", "random", ones=True)
run("tw_one_zeros", "zeros", ones=True)
run("random (again)", "random")
for bc, st in ((8, 2), (32, 1), (128, 1)):
    run("batch %dx%d" % (bc, st), "random", opts={"fft.batch_cols": bc, "fft.streams": st})
