"""Back-to-back row-sharded proves on W thread-ranks (one GPU), chunked exchange forced on: every rank's bytes must equal the single-rank
proof of the same seed, every time (races between the event-ordered exchange and the next chunk's LDE would show up here).
  python tools/sharded_soak.py [world=4] [log=15] [reps=20] [chunks=3]"""
import json, os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
log = int(sys.argv[2]) if len(sys.argv) > 2 else 15
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
os.environ["NX_DIST_CHUNKS"] = sys.argv[4] if len(sys.argv) > 4 else "3"
import numpy as np
import nexus_zkvm_amd as nz
from nexus_zkvm_amd.sharded import ThreadGroup

comps = [(log, 27, 120, 32), (log - 3, 3, 40, 8)]
cfg = nz.default_config(pow_bits=6)
be = nz.HipBackend(0)
refs = [be.prove_machine(comps, cfg, seed=500 + i) for i in range(reps)]
be.close()
group = ThreadGroup(world)
bad, errors = [], []


def run(rank):
    try:
        b = nz.HipBackend(0)
        comm = nz.make_comm(rank, world, group.comm(rank, b))
        for i in range(reps):
            w = b.prove_machine(comps, cfg, seed=500 + i, comm=comm)
            if not np.array_equal(w, refs[i]):
                bad.append((rank, i))
        b.close()
    except Exception as e:   # noqa: BLE001
        errors.append((rank, repr(e)))
        try: group.barrier.abort()
        except Exception: pass


th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
for t in th: t.start()
for t in th: t.join(timeout=1200)
print(json.dumps({"world": world, "log_rows": log, "proves_per_rank": reps, "chunks": os.environ["NX_DIST_CHUNKS"], "mismatches": bad, "errors": errors,
                  "ok": not bad and not errors}))
