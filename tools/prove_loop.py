"""Plain proves in a loop, NO statistics call (nx_prove_stats makes every stage end in a synchronisation): what `rocprofv3 --kernel-trace`
should watch when the question is where the GPU idles inside a TIMED prove (tools/kernel_sequence.py takes the last prove of the trace).
usage: prove_loop.py headline|v1|v1w|keccak|keccakw [--steps K] [--log-rows N]     prints ms per prove"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["headline", "v1", "v1w", "keccak", "keccakw"])
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--log-rows", type=int, default=22)
    a = ap.parse_args()
    import nexus_zkvm_amd as nz
    be = nz.HipBackend(0)
    n = a.log_rows
    if a.what == "headline":
        comps, cfg = [(n, 27, 347, 64)], nz.default_config(pow_bits=10)
    elif a.what in ("v1", "v1w"):
        main_c = (n, 27, 347, 1000, 2, nz.TUPLES_V1) if a.what == "v1w" else (n, 27, 347, 1000, 2)   # v1w: the reference's tuple widths
        comps = [main_c] + [(8 + k, 2, 6 + k, 4, 1) for k in range(8)]
        cfg = nz.default_config(pow_bits=10, log_constraint_degree=2)
    else:
        from keccak_shaped import keccak_shaped_components
        comps, cfg = keccak_shaped_components(0, 1000, 500, pairs=True, tuples=a.what == "keccakw"), nz.default_config(pow_bits=10)
    be.prove_machine(comps, cfg, seed=5); be.sync()
    t0 = time.perf_counter()
    for s in range(a.steps):
        be.prove_machine(comps, cfg, seed=100 + s)
    be.sync()
    print(json.dumps({"what": a.what, "log_rows": n, "steps": a.steps, "ms_per_prove": round(1e3 * (time.perf_counter() - t0) / a.steps, 3)}))
    be.close()


if __name__ == "__main__":
    main()
