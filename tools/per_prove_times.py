import sys, time, json, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import nexus_zkvm_amd as nz
be = nz.HipBackend(0)
comps, cfg = [(22, 27, 347, 64)], nz.default_config(pow_bits=10)
be.prove_machine(comps, cfg, seed=5); be.sync()
ts = []
for s in range(60):
    t0 = time.perf_counter(); be.prove_machine(comps, cfg, seed=100 + s); ts.append(round(1e3 * (time.perf_counter() - t0), 2))
print(json.dumps({"per_prove_ms": ts, "min": min(ts), "median": sorted(ts)[len(ts)//2], "max": max(ts), "mean": round(sum(ts)/len(ts), 3)}))
