"""Regenerates tests/golden/oracle_golden.json from the CPU oracle.

The reference has no golden vectors for this path and cannot be built here, so
these fixtures freeze the *oracle* (after tests/test_oracle_primitives.py's independent checks
pass); the GPU parity tests replay them through the HIP path.  Run: python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

P = O.P


def main():
    out = {"prove": [], "lde_commit": []}
    cases = [
        dict(comps=[[8, 3, 20, 6]], cfg=dict(pow_bits=8), seed=1, ad=[]),
        dict(comps=[[10, 27, 40, 8], [6, 2, 5, 4]], cfg=dict(pow_bits=10), seed=0xC0FFEE, ad=[1, 2, 3]),
        dict(comps=[[9, 4, 18, 4]], cfg=dict(pow_bits=5, log_constraint_degree=2), seed=2, ad=[]),
        dict(comps=[[8, 3, 20, 6], [8, 2, 3, 0], [5, 2, 2, 2]], cfg=dict(pow_bits=6, hash_mode=1, fri_alpha_mode=1), seed=3, ad=[7]),
    ]
    for c in cases:
        cfg = O.default_cfg(**c["cfg"])
        comps = [tuple(x) for x in c["comps"]]
        w = O.prove_synth(comps, cfg, seed=c["seed"], ad=bytes(c["ad"]))
        assert O.verify_synth(comps, cfg, w, ad=bytes(c["ad"])) is None
        c = dict(c)
        c["sha256"] = hashlib.sha256(w.tobytes()).hexdigest()
        c["root0"] = [int(x) for x in w[6:14]]
        c["n_words"] = int(len(w))
        out["prove"].append(c)
    for seed, logs in [(1, [6, 6, 6]), (2, [8, 5, 8, 6, 5]), (3, [10] * 20)]:
        rnd = np.random.default_rng(seed)
        cols = [rnd.integers(0, P, 1 << l, dtype=np.uint32) for l in logs]
        tw = O.Twiddles(max(logs))
        ldes = [tw.evaluate(tw.interpolate(c), int(np.log2(len(c))) + 1) for c in cols]
        out["lde_commit"].append(dict(seed=seed, logs=logs,
                                      root_std=[int(x) for x in O.merkle_commit(ldes, O.HASH_STD)],
                                      root_raw0=[int(x) for x in O.merkle_commit(ldes, O.HASH_RAW0)]))
    json.dump(out, open(os.path.join(HERE, "oracle_golden.json"), "w"), indent=1)
    print("wrote", os.path.join(HERE, "oracle_golden.json"))


if __name__ == "__main__":
    O.build_oracle()
    main()
