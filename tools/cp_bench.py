"""Times nx_eval_constraint_program on the synthetic machine's constraints at BASELINE config #3 width (prints one JSON line)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import nexus_zkvm_amd as nz
import nexus_zkvm_amd.air_program as ap
from test_air_program_cpu import denominators, synthetic_program
log = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_pre, n_main, n_inter = 27, 347, 64
be = nz.HipBackend(0)
e = log + 1
tw = be.precompute_twiddles(e)
ldes = []
for tree in range(3):
    for s in be.synth_fill_tree([(log, n_pre, n_main, n_inter)], tree, seed=5, inter_seed=99):
        ldes.append(be.lde(tw, s, 1)); s.free()
ptrs = [l.ptr.value + k * (4 << e) for l in ldes for k in range(l.n_cols)]
prog = synthetic_program(ap, n_pre, n_main, n_inter)
pw = np.random.default_rng(8).integers(0, nz.P, (prog.n_constraints, 4), dtype=np.uint32)
den = denominators(log, e)
acc = be.columns(4, e)
best = 1e9
for r in range(4):
    be.sync(); t0 = time.perf_counter()
    be.eval_constraint_program(prog, ptrs, pw, den, log, e, acc)
    be.sync(); best = min(best, time.perf_counter() - t0)
n_cols = len(ptrs)
import ctypes as C
t0 = time.perf_counter(); kern = be.compile_air(prog, n_cols); t_compile = time.perf_counter() - t0
acc2 = be.columns(4, e)
for a in (acc, acc2):
    be._chk(be.L.nx_memset_zero(be.ctx, a.ptr, C.c_size_t(4 << e)))
be.eval_constraint_program(prog, ptrs, pw, den, log, e, acc)
kern.eval(ptrs, pw, den, log, e, acc2)
same = bool(np.array_equal(acc.to_cpu(), acc2.to_cpu()))
jit = 1e9
for r in range(5):
    be.sync(); t0 = time.perf_counter()
    kern.eval(ptrs, pw, den, log, e, acc2)
    be.sync(); jit = min(jit, time.perf_counter() - t0)
print(json.dumps({"log_size": log, "columns": n_cols, "instructions": int(len(prog.instrs)), "registers": prog.n_regs, "constraints": prog.n_constraints,
                  "ms": best * 1e3, "jit_ms": jit * 1e3, "jit_equals_interpreter": same, "jit_compile_s": t_compile, "jit_column_GBs": n_cols * (4 << e) / jit / 1e9, "column_GBs": n_cols * (4 << e) / best / 1e9, "G_instr_rows_per_s": len(prog.instrs) * (1 << e) / best / 1e9}))
