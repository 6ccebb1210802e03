"""Overhead proxy for the row-sharded prove on a 1-GPU box: W thread-ranks (one context each) prove ONE 2^log-row machine on the SAME
GPU.  The total work equals the single-rank prove's, so  T(W ranks) - T(1 rank)  is what sharding adds apart from the wire: the pack /
unpack passes, the collectives' copies (device-to-device here instead of xGMI), the host round trips at every collective, the
replicated parts (tree tops, FRI tail, finalize_last) and the smaller launches.  Not a scaling measurement.
  python tools/thread_ranks_bench.py [log=22] [worlds=2,4,8] [options, e.g. air.quarter_domain=0 or -] [statement: headline | v1]
v1: the v1-shaped statement of bench.py (main component with the +2 bound and 250 logup columns, 8 extension components)."""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nexus_zkvm_amd as nz
from nexus_zkvm_amd.sharded import ThreadGroup

log = int(sys.argv[1]) if len(sys.argv) > 1 else 22
worlds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,4,8").split(",")]
OPTS = [kv.split("=") for kv in (sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else "").split(",") if kv]
STATEMENT = sys.argv[4] if len(sys.argv) > 4 else "headline"
if STATEMENT == "v1":
    comps = [(log, 27, 347, 1000, 2)] + [(8 + k, 2, 6 + k, 4, 1) for k in range(8)]
    cfg = nz.default_config(pow_bits=10, log_constraint_degree=2)
else:
    comps = [(log, 27, 347, 64)]
    cfg = nz.default_config(pow_bits=10)
REPS = 3


def backend():
    b = nz.HipBackend(0)
    for k, v in OPTS:
        b.set_option(k, int(v))
    return b


be = backend()
ref, best1 = None, 1e9
for rep in range(REPS + 1):
    be.sync(); t0 = time.perf_counter()
    ref = be.prove_machine(comps, cfg, seed=77)
    be.sync()
    if rep: best1 = min(best1, time.perf_counter() - t0)
_, st1 = be.prove_machine(comps, cfg, seed=77, want_stats=True)
be.close()
out = {"log_rows": log, "statement": STATEMENT, "options": dict((k, int(v)) for k, v in OPTS), "single_rank_ms": best1 * 1e3, "worlds": {}}


class Counting:
    """counts the collectives one rank issues during a prove"""
    def __init__(self, impl): self.impl, self.calls, self.on = impl, {}, False
    def __getattr__(self, name):
        f = getattr(self.impl, name)
        def g(*a):
            if self.on:
                k = name + (":%d" % len(a[0]) if name == "allgather" else ":%d" % a[1] if name == "allgather_dev" else "")
                self.calls[k] = self.calls.get(k, 0) + 1
            return f(*a)
        return g


NATIVE = os.environ.get("NX_RANKS_TRANSPORT", "native") == "native"     # "python": sharded.ThreadGroup behind the callback trampoline (rounds 1-3)


def run_world(world):
    group = nz.LocalGroup(world) if NATIVE else ThreadGroup(world)
    times, stats, same, errors, counts = [1e9] * world, [None] * world, [False] * world, [], {}
    start = threading.Barrier(world)

    def run(rank):
        try:
            b = backend()
            if NATIVE:
                cnt, comm = None, b.local_comm(group, rank)
            else:
                cnt = Counting(group.comm(rank, b))
                comm = nz.make_comm(rank, world, cnt)
            for rep in range(REPS + 1):
                b.sync(); start.wait(); t0 = time.perf_counter()
                w = b.prove_machine(comps, cfg, seed=77, comm=comm)
                b.sync(); start.wait()
                if rep: times[rank] = min(times[rank], time.perf_counter() - t0)
            same[rank] = bool(np.array_equal(w, ref))
            if cnt: cnt.on = True
            stats[rank] = b.prove_machine(comps, cfg, seed=77, comm=comm, want_stats=True)[1]
            if rank == 0 and cnt: counts.update(cnt.calls)
            if NATIVE: b.free_local_comm(comm)
            b.close()
        except Exception as e:   # noqa: BLE001
            errors.append((rank, repr(e)))
            try: start.abort()
            except Exception: pass
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join(timeout=900)
    if errors: return {"errors": errors}
    s = stats[0]
    return {"transport": "native (csrc/comm_local.hip)" if NATIVE else "python (sharded.ThreadGroup)", "ms": max(times) * 1e3, "same_proof_bytes": all(same), "rank0_stages_ms": {k: round(v, 3) for k, v in s.items() if isinstance(v, float) and k not in ("comm_ms",)},
            "rank0_comm_ms": s.get("comm_ms"), "rank0_comm_bytes": s.get("comm_bytes"),
            "rank0_collectives": counts, "rank0_collective_calls": sum(counts.values()),
            # counted by the library itself (nx_prove_stats, any transport): each is also a host synchronisation of the rank's stream
            "rank0_n_alltoallv": s.get("n_alltoallv"), "rank0_n_allgather_dev": s.get("n_allgather_dev"), "rank0_n_allgather_host": s.get("n_allgather_host")}


for w in worlds:
    out["worlds"][str(w)] = run_world(w)
out["single_rank_stages_ms"] = {k: round(v, 3) for k, v in st1.items() if isinstance(v, float)}
print(json.dumps(out))
