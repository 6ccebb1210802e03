"""A wide logup-style AIR (the shape of BASELINE config #5: few main columns per lookup, many secure interaction columns) through
the whole device pipeline: main trace in HBM -> lookup elements from the session channel -> K interaction columns by nx_logup_col +
nx_logup_finalize_last written straight into the session's tree -> recorded constraints (secure-field arithmetic) compiled by
hiprtc -> nx_prover_prove.  Prints one JSON line with the stage times."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import nexus_zkvm_amd as nz
import nexus_zkvm_amd.air_program as ap

P = nz.P
log = int(sys.argv[1]) if len(sys.argv) > 1 else 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32          # logup columns; 2 main columns each
n = 1 << log
be = nz.HipBackend(0)
cfg = nz.default_config()
rng = np.random.default_rng(1)
main_host = rng.integers(0, P, (2 * K, n), dtype=np.uint32)


def qmul(a, b):    # QM31 product on the host (tiny): via the library-free formula
    def cm(x, y): return ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)
    a = [int(v) for v in a]; b = [int(v) for v in b]
    aa, bb = cm(a[:2], b[:2]), cm(a[2:], b[2:])
    ab, ba = cm(a[:2], b[2:]), cm(a[2:], b[:2])
    r = ((2 * bb[0] - bb[1]) % P, (2 * bb[1] + bb[0]) % P)
    return np.array([(aa[0] + r[0]) % P, (aa[1] + r[1]) % P, (ab[0] + ba[0]) % P, (ab[1] + ba[1]) % P], np.uint32)


def build_component(z, alpha, shifts):
    pb = ap.ProgramBuilder()
    ze, al = pb.econst(z), pb.econst(alpha)
    for j in range(K):
        (a,) = pb.next_trace_mask(2 * j)
        (b,) = pb.next_trace_mask(2 * j + 1)
        s_prev, s_cur = pb.next_secure_mask(2 * K + 4 * j, (-1, 0))
        pb.add_constraint((s_cur - s_prev + pb.econst(shifts[j])) * (ze - a - al * b) - 1)
    cols = [(1, k) for k in range(2 * K)] + [(2, k) for k in range(4 * K)] + [(0, 0)]
    prog = pb.build()
    return ap.Component(log, prog, cols, [prog.masks.get(k, [0]) for k in range(len(cols))])


def run(kern=None):
    t = {}
    be.sync(); t0 = time.perf_counter()
    s = be.prover_session(cfg, log)
    s.mix_u64(log)
    p0 = s.tree_begin([log]); be._chk(be.L.nx_memset_zero(be.ctx, C.c_void_p(p0[0]), C.c_size_t(n))); s.tree_commit()
    mp = s.tree_begin([log] * (2 * K))
    mainv = nz.DeviceColumns.view(be, mp[0], 2 * K, log)
    be._chk(be.L.nx_copy(be.ctx, mainv.ptr, d_main.ptr, C.c_size_t(2 * K * n)))      # "trace generation": the main trace is already in HBM
    s.tree_commit()
    be.sync(); t["commit_pre_main"] = time.perf_counter() - t0; t1 = time.perf_counter()
    z, alpha = s.draw_felt(), s.draw_felt()
    ip = s.tree_begin([log] * (4 * K))
    ninv = [pow(n, P - 2, P), 0, 0, 0]
    Ss = []
    for j in range(K):
        S = nz.DeviceColumns.view(be, ip[4 * j], 4, log)
        tup = nz.DeviceColumns.view(be, d_main.ptr.value + 2 * j * 4 * n, 2, log)
        be.logup_col({"tuple": tup, "alphas": np.array([[1, 0, 0, 0], alpha], np.uint32), "z": z, "scale": (P - 1, 0, 0, 0)}, out=S)
        Ss.append(S)
    claimed = list(be.logup_finalize_last_batch(Ss))        # one call: three launches, one copy
    shifts = [qmul(c, ninv) for c in claimed]
    be.sync(); t["interaction_trace"] = time.perf_counter() - t1; t2 = time.perf_counter()
    s.mix_felts(np.stack(claimed))
    s.tree_commit()
    comp = build_component(z, alpha, shifts)
    be.sync(); t["commit_inter_and_record"] = time.perf_counter() - t2; t3 = time.perf_counter()
    kern = kern or be.compile_air(comp.program, len(comp.cols))
    t["jit_compile"] = time.perf_counter() - t3; t4 = time.perf_counter()
    words, st = s.prove([comp], kernels=[kern], want_stats=True)
    be.sync(); t["prove"] = time.perf_counter() - t4
    t["total"] = time.perf_counter() - t0
    s.close()
    return t, st, len(words), kern, comp


d_main = be.columns_from_host(main_host)
t, st, nw, kern, comp = run()
t, st, nw, _, _ = run(kern)       # second pass: kernel already compiled (the lookup elements differ only in value)
print(json.dumps({"log_size": log, "logup_columns": K, "main_columns": 2 * K, "interaction_columns": 4 * K, "instructions": int(len(comp.program.instrs)),
                  "registers": comp.program.n_regs, "ms": {k: round(v * 1e3, 3) for k, v in t.items()},
                  "prove_stages_ms": {k: round(v, 3) for k, v in st.items() if k in ("commit", "composition", "oods", "quotients", "fri", "total")}, "proof_words": nw}))
