"""Leaf hashing of 347 columns x 2^23 rows, standalone: nx_merkle_leaf_chain (the prover's TreeBuilder path) against nx_merkle_commit
(merkle_layer_kernel on the leaf level + the inner levels), after an idle gap and right after an LDE (the prover's situation)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nexus_zkvm_amd as nz
be = nz.HipBackend(0)
log, ncols = 22, 347
tw = be.precompute_twiddles(log)
cols = be.synth_fill_tree([(log, 2, ncols, 0)], 1, seed=3)[0]
out = be.columns(ncols, log + 1)
state = be.columns(8, log + 1)
def lde():
    be._chk(be.L.nx_lde_batch(be.ctx, tw.h, cols.col_ptrs(), ncols, log, 1, out.col_ptrs()))
def leaf():
    be.merkle_leaf_chain(out, 0, ncols, None, state.ptr.value)
def commit():
    return be.merkle_commit([out])
def timeit(f, pre=None):
    best = []
    for r in range(4):
        if pre: pre()
        else: be.sync(); time.sleep(0.05)
        t0 = None
        be.sync() if pre is None else None
        t0 = time.perf_counter(); f(); be.sync(); best.append(time.perf_counter() - t0)
    return [round(x * 1e3, 3) for x in best]
lde(); be.sync()
n_leaf = (1 << (log + 1)) * 22
print(json.dumps({"leaf_chain_after_idle_ms": timeit(leaf), "commit_after_idle_ms": timeit(commit)}))
# right behind an LDE (no sync in between: the Merkle kernels start on a chip that ran at the power cap)
def after_lde(f):
    res = []
    for r in range(4):
        be.sync(); t0 = time.perf_counter(); lde(); be.sync(); t_l = time.perf_counter() - t0
        be.sync(); t0 = time.perf_counter(); lde(); f(); be.sync(); res.append(round((time.perf_counter() - t0 - t_l) * 1e3, 3))
    return res
print(json.dumps({"leaf_chain_behind_lde_ms": after_lde(leaf), "commit_behind_lde_ms": after_lde(commit), "leaf_compressions": n_leaf}))
