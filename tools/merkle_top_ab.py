"""Same-box A/B of where the one-block tree top starts ("merkle.top", "merkle.subtree"): the headline machine at 2^16 / 2^18 / 2^22 rows, option sets
interleaved, median ms per prove; the proofs of all sets must be the same bytes.  usage: merkle_top_ab.py [rounds=5]"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nexus_zkvm_amd as nz
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
be = nz.HipBackend(0)
cfg = nz.default_config(pow_bits=10)
SETS = [("top%d_sub%d" % (t, t + 7), {"merkle.top": t, "merkle.subtree": t + 7}) for t in (int(x) for x in os.environ.get("NX_AB_TOPS", "10,8,7,6,5,4,3").split(","))]
if os.environ.get("NX_AB_TAILS"):       # second sweep: the size from which the FRI tail is one launch ("fri.tail"), at the default top
    SETS = [("tail%d" % t, {"fri.tail": t}) for t in (int(x) for x in os.environ["NX_AB_TAILS"].split(","))]
for log, reps in ((16, 40), (18, 30), (22, 6)):
    comps = [(log, 27, 347, 64)]
    be.prove_machine(comps, cfg, seed=5)
    samples = {name: [] for name, _ in SETS}; digests = set()
    for r in range(rounds):
        for name, opts in SETS:
            for k, v in opts.items():
                be.set_option(k, v)
            w = be.prove_machine(comps, cfg, seed=7); be.sync()
            digests.add(hashlib.sha256(w.tobytes()).hexdigest())
            t0 = time.perf_counter()
            for s in range(reps):
                be.prove_machine(comps, cfg, seed=100 + s)
            be.sync(); samples[name].append(1e3 * (time.perf_counter() - t0) / reps)
    print(json.dumps({"log_rows": log, "same_proof_bytes": len(digests) == 1, "median_ms": {n: round(sorted(v)[len(v) // 2], 4) for n, v in samples.items()},
                      "min_ms": {n: round(min(v), 4) for n, v in samples.items()}}), flush=True)
be.close()
