"""A prover2-shaped statement (reference prover2: ~55 components of different sizes): many components, few columns each, sizes from
2^10 to 2^20.  Times one prove and prints the stage split (launch-bound paths show up here, not in the one-component headline)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nexus_zkvm_amd as nz
be = nz.HipBackend(0)
cfg = nz.default_config()
comps = [(20, 2, 60, 40)] * 2 + [(18, 2, 40, 24)] * 6 + [(16, 2, 30, 16)] * 10 + [(14, 2, 24, 12)] * 12 + [(12, 2, 20, 8)] * 14 + [(10, 2, 12, 8)] * 11
be.prove(comps, cfg, seed=3)
best, st_best = 1e9, None
for _ in range(4):
    be.sync(); t0 = time.perf_counter(); w, st = be.prove(comps, cfg, seed=3, want_stats=True); be.sync(); dt = time.perf_counter() - t0
    if dt < best: best, st_best = dt, st
cols = sum(a + b + c for _, a, b, c in comps)
cells = sum((a + b + c) << l for l, a, b, c in comps)
print(json.dumps({"components": len(comps), "columns": cols, "trace_cells": cells, "ms": best * 1e3, "cells_per_s": cells / best,
                  "stages_ms": {k: round(v, 3) for k, v in st_best.items() if k in ("trace_gen", "commit", "composition", "oods", "quotients", "fri", "pow", "decommit", "total")}}))
