"""nx_logup_program (the interaction trace from the recorded AIR's relation entries: hiprtc-compiled fraction program) against
nx_logup_cols (the hand-fed descriptor form nx_prove_machine uses) on the SAME fractions: the v1-shaped machine's logup columns —
fraction f: 1 / (main[a_f] - z) or (f odd) 1 / (main[a_f] + alpha main[b_f] - z), numerator -main[m_f] when f % 3 == 2.
  python tools/logup_program_bench.py [log=22] [n_main=347] [n_logup=250] [pairs=0]
Prints one JSON line: ms per call of both, GB/s of the written columns, and whether the two outputs are equal word for word."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nexus_zkvm_amd as nz
import nexus_zkvm_amd.air_program as ap

log = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n_main = int(sys.argv[2]) if len(sys.argv) > 2 else 347
L = int(sys.argv[3]) if len(sys.argv) > 3 else 250
pairs = len(sys.argv) > 4 and sys.argv[4] == "1"
P = (1 << 31) - 1
be = nz.HipBackend(0)
main = be.synth_fill_tree([(log, 2, n_main, 0)], 1, seed=3)[0]
ptr = lambda k: main.ptr.value + k * (4 << log)
rng = np.random.default_rng(1)
z, alpha = rng.integers(0, P, 4, dtype=np.uint32), rng.integers(0, P, 4, dtype=np.uint32)
F = 2 * L if pairs else L

pb = ap.ProgramBuilder()
cols = [pb.next_trace_mask(k)[0] for k in range(n_main)]
rel = pb.relation(z, alpha, 2)
for f in range(F):
    a, b, m = (3 + 7 * f) % n_main, (5 + 11 * f) % n_main, (2 + 13 * f) % n_main
    pb.add_to_relation(rel, -cols[m] if f % 3 == 2 else 1, [cols[a], cols[b]] if f & 1 else [cols[a]])
(pb.finalize_logup_in_pairs if pairs else pb.finalize_logup)(n_main, (0, 0, 0, 0))
frac = pb.build_logup()
ptrs = [ptr(k) for k in range(n_main)] + [None] * (4 * L)
t0 = time.perf_counter(); out = be.logup_program(frac, ptrs, log); be.sync(); first = time.perf_counter() - t0
for o in out[:-1]:
    o.free()
keep_last = out[-1].to_cpu()
out[-1].free()
best = 1e9
for _ in range(3):
    be.sync(); t0 = time.perf_counter()
    out = be.logup_program(frac, ptrs, log); be.sync()
    best = min(best, time.perf_counter() - t0)
    for o in out:
        o.free()

apw = np.stack([np.array([1, 0, 0, 0], np.uint32), alpha])
one, minus = (1, 0, 0, 0), (P - 1, 0, 0, 0)
fr = []
for f in range(F):
    a, b, m = (3 + 7 * f) % n_main, (5 + 11 * f) % n_main, (2 + 13 * f) % n_main
    tup = nz.DeviceColumns.view(be, ptr(a), 1, log) if not f & 1 else None
    d = dict(alphas=apw[:2 if f & 1 else 1], z=z, scale=minus if f % 3 == 2 else one)
    if f & 1:       # two tuple columns that are not contiguous: a 2-column slab copy
        t2 = be.columns(2, log)
        be._chk(be.L.nx_copy(be.ctx, t2.ptr, __import__("ctypes").c_void_p(ptr(a)), __import__("ctypes").c_size_t(1 << log)))
        be._chk(be.L.nx_copy(be.ctx, __import__("ctypes").c_void_p(t2.ptr.value + (4 << log)), __import__("ctypes").c_void_p(ptr(b)), __import__("ctypes").c_size_t(1 << log)))
        tup = t2
    d["tuple"] = tup
    if f % 3 == 2:
        d["mult"] = nz.DeviceColumns.view(be, ptr(m), 1, log)
    fr.append(d)
best2 = 1e9
for _ in range(3):
    be.sync(); t0 = time.perf_counter()
    o2 = be.logup_cols_batched(fr, None, L) if pairs else be.logup_cols(fr)
    be.sync(); best2 = min(best2, time.perf_counter() - t0)
    last2 = o2[-1].to_cpu()
    for o in o2:
        o.free()
print(json.dumps({"log_rows": log, "n_main": n_main, "logup_columns": L, "fractions": F, "n_instr": int(len(frac.instrs)), "first_call_ms_with_hiprtc": round(first * 1e3, 1),
                  "logup_program_ms": round(best * 1e3, 3), "logup_cols_ms": round(best2 * 1e3, 3), "written_GB": round(16 * L * (1 << log) / 1e9, 2),
                  "program_write_GBs": round(16 * L * (1 << log) / best / 1e9, 1), "last_column_equal": bool(np.array_equal(keep_last, last2))}))
