"""Throughput mode: K independent proofs in flight on one GPU (one context + streams per host thread).  The protocol's sequential
points (root -> channel -> next stage) leave the GPU idle for ~2 ms of a 44 ms prove; a second proof fills them.  Prints one JSON line."""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nexus_zkvm_amd as nz
log = int(sys.argv[1]) if len(sys.argv) > 1 else 22
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
comps = [(log, 27, 347, 64)]
cfg = nz.default_config()
bes = [nz.HipBackend(0) for _ in range(K)]
for b in bes: b.prove(comps, cfg, seed=1)          # warm-up (allocator caches, twiddles)
def single():
    t0 = time.perf_counter()
    for _ in range(reps): bes[0].prove(comps, cfg, seed=1)
    return (time.perf_counter() - t0) / reps
t1 = single()
bar = threading.Barrier(K + 1)
def worker(b):
    bar.wait()
    for _ in range(reps): b.prove(comps, cfg, seed=1)
    bar.wait()
th = [threading.Thread(target=worker, args=(b,)) for b in bes]
for t in th: t.start()
bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
for t in th: t.join()
print(json.dumps({"log_size": log, "proofs_in_flight": K, "single_ms": t1 * 1e3, "concurrent_ms_per_proof": dt / (reps * K) * 1e3,
                  "single_cycles_per_s": (1 << log) / t1, "concurrent_cycles_per_s": (1 << log) * reps * K / dt}))
