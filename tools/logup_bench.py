"""Times the logup interaction-trace kernels at 2^log rows (prints one JSON line): combine over a t-column tuple, finalize_col
(one fraction / two merged fractions on top of a previous column), finalize_last."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nexus_zkvm_amd as nz
log = int(sys.argv[1]) if len(sys.argv) > 1 else 22
be = nz.HipBackend(0)
rng = np.random.default_rng(1)
tup = be.synth_fill_tree([(log, 2, 8, 0)], 1, seed=3)[0]
alphas = rng.integers(0, nz.P, (8, 4), dtype=np.uint32); z = rng.integers(0, nz.P, 4, dtype=np.uint32)
def timed(f, reps=4):
    best = 1e9
    for _ in range(reps):
        be.sync(); t0 = time.perf_counter(); r = f(); be.sync(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
t_comb, den = timed(lambda: be.logup_combine(tup, alphas, z))
mult = be.synth_fill_tree([(log, 2, 2, 0)], 1, seed=9)[0]
m0 = nz.DeviceColumns.__new__(nz.DeviceColumns); m0.be, m0.n_cols, m0.log_size, m0.ptr = be, 1, log, mult.ptr
t_one, col0 = timed(lambda: be.logup_finalize_col(den))
t_two, col1 = timed(lambda: be.logup_finalize_col(den, mult_a=m0, den_b=den, mult_b=m0, prev=col0))
one = be.columns(1, log); nz.DeviceColumns.upload  # 1-column tuple = a range-check limb
limb = nz.DeviceColumns.__new__(nz.DeviceColumns); limb.be, limb.n_cols, limb.log_size, limb.ptr = be, 1, log, tup.ptr
t_f1, f0 = timed(lambda: be.logup_col(dict(tuple=limb, alphas=alphas[:1], z=z)))
t_f2, f1 = timed(lambda: be.logup_col(dict(tuple=limb, alphas=alphas[:1], z=z, mult=m0), dict(tuple=tup, alphas=alphas, z=z, mult=m0), prev=f0))
limb.ptr = nz.C.c_void_p()
m0.ptr = nz.C.c_void_p()   # the alias must not free the slab
t_last, _ = timed(lambda: be.logup_finalize_last(col1), reps=2)
n = 1 << log
print(json.dumps({"log_size": log, "combine_8col_ms": t_comb, "finalize_col_1frac_ms": t_one, "finalize_col_2frac_prev_ms": t_two, "finalize_last_ms": t_last, "fused_col_1limb_ms": t_f1, "fused_col_2frac_prev_ms": t_f2,
                  "finalize_col_Grows_per_s": n / t_two / 1e6}))
