"""The headline prove (BASELINE config #1 shape: 2^log rows x 438 columns) through the generic prover session (nx_prover_* with
the AIR as a RECORDED program compiled by hiprtc) next to the hand-written nx_prove_synth.  Same proof bytes; prints one JSON line."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import nexus_zkvm_amd as nz
import nexus_zkvm_amd.air_program as ap
from test_air_program_cpu import synthetic_program

log = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n_pre, n_main, n_inter = 27, 347, 64
comps = [(log, n_pre, n_main, n_inter)]
be = nz.HipBackend(0)
cfg = nz.default_config()
prog = synthetic_program(ap, n_pre, n_main, n_inter)
cols = [(0, k) for k in range(n_pre)] + [(1, k) for k in range(n_main)] + [(2, k) for k in range(n_inter)]
comp = ap.Component(log, prog, cols)
kern = be.compile_air(prog, len(cols))
carr = be._comps(comps)


def fill(session, tree, n, inter_seed=0):
    ptrs = session.tree_begin([log] * n)
    arr = (C.c_void_p * max(1, n))(*ptrs)
    be._chk(be.L.nx_synth_fill_tree(be.ctx, carr, 1, tree, C.c_uint64(1), C.c_uint64(inter_seed), arr))
    return session.tree_commit()


def session_prove():
    s = be.prover_session(cfg, log)
    s.mix_u64(log)
    fill(s, 0, n_pre); fill(s, 1, n_main)
    z = s.draw_felt()
    inter_seed = (int(z[0]) << 32) ^ int(z[1]) ^ (int(z[2]) << 16) ^ (int(z[3]) << 48)
    s.mix_felts(np.zeros(4, np.uint32))
    fill(s, 2, n_inter, inter_seed)
    words, st = s.prove([comp], kernels=[kern], want_stats=True)
    s.close()
    return words, st


ref = be.prove(comps, cfg, seed=1)
w, _ = session_prove()
same = bool(np.array_equal(w, ref))
best_s, best_h, st_best = 1e9, 1e9, None
for _ in range(4):
    be.sync(); t0 = time.perf_counter(); w, st = session_prove(); be.sync(); dt = time.perf_counter() - t0
    if dt < best_s: best_s, st_best = dt, st
    be.sync(); t0 = time.perf_counter(); be.prove(comps, cfg, seed=1); be.sync(); best_h = min(best_h, time.perf_counter() - t0)
print(json.dumps({"log_size": log, "columns": len(cols), "same_proof_bytes": same, "session_ms": best_s * 1e3, "hand_written_ms": best_h * 1e3,
                  "session_composition_ms": st_best["composition"], "cycles_per_s_session": (1 << log) / best_s}))
