"""Where the LDE kernels' wave cycles go: SQ counters of the fft13 / lde_mid / leaf-hash kernels from rocprofv3 --pmc passes (8 SQ
counters per pass, MI355X_MICROARCH.md §rocprofv3 PMC slots; --kernel-trace only).  Workload: nx_lde_batch of `cols` columns at 2^log
rows followed by a Merkle commit of the result.
  python tools/pmc_sq.py [--log 22] [--cols 64] [--out profiles/r02_fft_sq_counters.json]"""
import argparse, csv, glob, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOAD = r'''
import sys; sys.path.insert(0, %(root)r)
import numpy as np
import nexus_zkvm_amd as nz
be = nz.HipBackend(0)
for kv in %(opts)r.split(","):
    if kv: be.set_option(kv.split("=")[0], int(kv.split("=")[1]))
log, ncols = %(log)d, %(cols)d
tw = be.precompute_twiddles(log)
cols = be.synth_fill_tree([(log, 2, ncols, 0)], 1, seed=3)[0]
out = be.columns(ncols, log + 1)
be.sync()
be._chk(be.L.nx_lde_batch(be.ctx, tw.h, cols.col_ptrs(), ncols, log, 1, out.col_ptrs()))
be.sync()
t = be.merkle_commit([out])
be.sync()
'''
PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"],
    ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_WAVES"],
    ["SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_LDS_DATA_FIFO_FULL", "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_WR_TA_DATA_FIFO_FULL", "SQ_THREAD_CYCLES_VALU"],
]


def run_pass(idx, counters, log, cols, workdir, opts=""):
    script = os.path.join(workdir, "wl.py")
    open(script, "w").write(WORKLOAD % {"root": ROOT, "log": log, "cols": cols, "opts": opts})
    out = os.path.join(workdir, "p%d" % idx)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.check_call(["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable, script],
                          cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        e = agg.setdefault(k, {})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log", type=int, default=22)
    ap.add_argument("--cols", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_fft_sq_counters.json"))
    ap.add_argument("--opts", default="", help="nx_ctx_set_option settings, name=value,name=value (e.g. fft.pipe=0)")
    ap.add_argument("--passes", default="0,1,2")
    a = ap.parse_args()
    res = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as wd:
        for i, c in enumerate(PASSES):
            if str(i) not in a.passes.split(","): continue
            try:
                for k, v in run_pass(i, c, a.log, a.cols, wd, a.opts).items():
                    res.setdefault(k, {}).update(v)
            except Exception as e:   # noqa: BLE001
                res.setdefault("_errors", {})["pass%d" % i] = repr(e)
    keep = {k: v for k, v in res.items() if any(q in k for q in ("fft13", "lde_mid", "pipe_", "merkle", "_errors"))}
    for k, v in keep.items():
        wc = v.get("SQ_WAVE_CYCLES")
        if wc:
            v["frac_of_wave_cycles"] = {n: round(v[n] / wc, 4) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                                                                        "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS") if n in v}
        if "SQ_LDS_IDX_ACTIVE" in v and v["SQ_LDS_IDX_ACTIVE"]:
            v["lds_conflict_cycles_per_lds_cycle"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"], 4)
    out = {"workload": "nx_lde_batch %d columns 2^%d -> 2^%d, then nx_merkle_commit" % (a.cols, a.log, a.log + 1), "options": a.opts, "kernels": keep}
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
