"""HBM traffic of the Circle-FFT (LDE) kernels from rocprofv3 PMC counters -> profiles/fft_traffic.json.

Runs on the GPU box.  Two separate --pmc passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2:
MI355X_MICROARCH.md §rocprofv3 PMC slots), each only with --kernel-trace.  The counter units are calibrated in
the same pass on a kernel with a known byte count in the same access pattern (nx_copy: 16 B per lane streaming
read + write), as MI355X_MICROARCH.md §HBM prescribes (gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read).

  python tools/pmc_traffic.py [--log 22] [--cols 64] [--out profiles/fft_traffic.json]
"""
import argparse, csv, glob, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOAD = r'''
import sys; sys.path.insert(0, %(root)r)
import nexus_zkvm_amd as nz
be = nz.HipBackend(0)
log, ncols = %(log)d, %(cols)d
tw = be.precompute_twiddles(log)
cols = be.synth_fill_tree([(log, 2, ncols, 0)], 1, seed=3)[0]
out = be.columns(ncols, log + 1)
be.sync()
c2 = be.clone_columns(out)          # calibration: 16 B/lane copy of ncols * 2^(log+1) words
be.sync()
be._chk(be.L.nx_lde_batch(be.ctx, tw.h, cols.col_ptrs(), ncols, log, 1, out.col_ptrs()))
be.sync()
'''


def run_pass(counter, log, cols, workdir):
    script = os.path.join(workdir, "wl.py")
    open(script, "w").write(WORKLOAD % {"root": ROOT, "log": log, "cols": cols})
    out = os.path.join(workdir, counter)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.check_call(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--",
                           sys.executable, script], cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        e = agg.setdefault(k, [0.0, 0])
        e[0] += float(r["Counter_Value"]); e[1] += 1
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log", type=int, default=22)
    ap.add_argument("--cols", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "fft_traffic.json"))
    a = ap.parse_args()
    res = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as wd:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            res[counter] = run_pass(counter, a.log, a.cols, wd)
    copy_bytes = a.cols * (4 << (a.log + 1))
    def pick(agg, pat):
        pats = ("fft", "lde_mid") if pat == "fft" else (pat,)     # the LDE's kernels: fft13_kernel passes + the fused middle launch
        return {k: v for k, v in agg.items() if any(q in k for q in pats)}
    cal = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        ck = pick(res[counter], "copy_kernel")
        raw = sum(v[0] for v in ck.values()) * 1024.0
        cal[counter] = copy_bytes / raw if raw > 0 else None
    fft_fetch = pick(res["FETCH_SIZE"], "fft"); fft_write = pick(res["WRITE_SIZE"], "fft")
    raw_fetch = sum(v[0] for v in fft_fetch.values()) * 1024.0
    raw_write = sum(v[0] for v in fft_write.values()) * 1024.0
    launches = sum(v[1] for v in fft_fetch.values())
    hbm = raw_fetch * (cal["FETCH_SIZE"] or 2.0) + raw_write * (cal["WRITE_SIZE"] or 1.0)
    alg = a.cols * (16 << a.log)
    out = {
        "workload": "nx_lde_batch: %d columns, 2^%d -> 2^%d (iFFT + FFT), blow-up 2" % (a.cols, a.log, a.log + 1),
        "kernels": sorted(set(list(fft_fetch) + list(fft_write))),
        "fft_kernel_launches": launches,
        "raw_FETCH_SIZE_bytes": raw_fetch, "raw_WRITE_SIZE_bytes": raw_write,
        "calibration": {"kernel": "nx::copy_kernel (16 B/lane)", "known_bytes_each_way": copy_bytes,
                        "fetch_scale": cal["FETCH_SIZE"], "write_scale": cal["WRITE_SIZE"]},
        "hbm_bytes_total": hbm, "algorithmic_bytes_total": alg,
        "hbm_bytes_per_column": hbm / a.cols, "algorithmic_bytes_per_column": alg / a.cols,
        "traffic_over_algorithmic": hbm / alg,
    }
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
