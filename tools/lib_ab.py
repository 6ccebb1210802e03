"""Same-box A/B of two (or more) builds of the library on the headline prove: each build runs in its own process (NX_LIB), the builds alternate for `rounds`
rounds; per build: ms per plain prove (median over rounds of a 10-prove loop), the median of every stage of 5 statistics proves, and the proof digest (all
builds must agree).  usage: lib_ab.py rounds name=path [name=path ...]   (path relative to nexus-zkvm_amd/; `default` = libnexus_hip.so)
                     lib_ab.py --child        (internal)"""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import nexus_zkvm_amd as nz
    be = nz.HipBackend(0)
    what = os.environ.get("NX_AB_WHAT", "headline")
    comps, cfg = [(22, 27, 347, 64)], nz.default_config(pow_bits=10)
    if what.startswith("log"):            # the headline machine at 2^N rows: NX_AB_WHAT=log16
        comps = [(int(what[3:]), 27, 347, 64)]
    if what in ("keccak", "keccakw"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from keccak_shaped import keccak_shaped_components
        comps = keccak_shaped_components(0, 1000, 500, pairs=True, tuples=what == "keccakw")
    if what == "v1":
        comps, cfg = [(22, 27, 347, 1000, 2)] + [(8 + k, 2, 6 + k, 4, 1) for k in range(8)], nz.default_config(pow_bits=10, log_constraint_degree=2)
    w = be.prove_machine(comps, cfg, seed=5); be.sync()
    n = 10 if what == "headline" else 3 if what == "v1" else 6 if what.startswith("keccak") else 100
    t0 = time.perf_counter()
    for s in range(n):
        be.prove_machine(comps, cfg, seed=100 + s)
    be.sync(); ms = 1e3 * (time.perf_counter() - t0) / n
    st = [be.prove_machine(comps, cfg, seed=200 + s, want_stats=True)[1] for s in range(2 if what == "v1" else 5)]
    keys = ("commit", "interaction", "composition", "oods", "quotients", "fri", "lde_kernel_ms", "merkle_kernel_ms")
    print(json.dumps({"ms": ms, "digest": hashlib.sha256(w.tobytes()).hexdigest(), "stages": {k: sorted(x[k] for x in st)[len(st) // 2] for k in keys}}))
    sys.exit(0)
rounds = int(sys.argv[1]); builds = [a.split("=", 1) for a in sys.argv[2:]]
res = {n: [] for n, _ in builds}
for r in range(rounds):
    for name, path in builds:
        env = dict(os.environ)
        if path.startswith("env:"):           # name=env:VAR=VALUE[,VAR=VALUE]: the default library under other environment defaults (context options)
            for kv in path[4:].split(","):
                k, v = kv.split("=", 1); env[k] = v
        elif name != "default":
            env["NX_LIB"] = os.path.join(ROOT, "nexus-zkvm_amd", path)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        res[name].append(json.loads(out))
med = lambda v: sorted(v)[len(v) // 2]
digests = {x["digest"] for v in res.values() for x in v}
for name, v in res.items():
    print(json.dumps({"build": name, "what": os.environ.get("NX_AB_WHAT", "headline"), "rounds": rounds, "ms_per_prove_median": round(med([x["ms"] for x in v]), 3), "ms_per_prove_min": round(min(x["ms"] for x in v), 3),
                      "stages_median": {k: round(med([x["stages"][k] for x in v]), 3) for k in v[0]["stages"]}, "same_proof_bytes": len(digests) == 1}))
