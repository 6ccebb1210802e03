"""BASELINE config #5 at the size SURVEY.md §8(d) states — the keccak-precompile guest's AIR shape — proved on ONE GPU through
nx_prove_machine (real logup interaction trace on the device, recorded AIR, hiprtc):
  * two keccak round components: 2^14 permutation instances x 16 resp. 8 rounds per row block -> log sizes 18 and 17
    (reference prover/src/extensions/keccak/mod.rs:15-24); a lane is 8 byte columns (round/constants.rs:45) -> ~10^3 main columns;
    4 logup columns per lane-level lookup (round/interaction_trace.rs:51-70,113-114,158-159) x ~130 lookups per round
    (round/mod.rs:11-51) -> ~0.5 k logup columns = ~2 k interaction base columns per round component;
  * the XOR and NOT-AND tables at log 12 (keccak/bitwise_table/mod.rs:108), the rotate table at log 11 (keccak/bit_rotate/mod.rs:57);
  * every extension's constraint-degree bound is +1 (keccak/round/eval.rs:28-30).
usage: keccak_shaped.py [--shift S] [--steps K] [--check]     --shift: every log size reduced by S bits (--check: against the oracle)
Prints one JSON line: ms per prove, stage split, LDE / Merkle kernel time."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


PAIRS, ODD, TABLE = 1, 2, 4     # nx_component_spec.logup_mode (include/nexus_hip.h NX_LOGUP_*)


TUPLES_KECCAK = 2 << 4          # NX_LOGUP_TUPLES(NX_TUPLES_KECCAK)


def keccak_shaped_components(shift=0, n_main=1000, n_logup=500, pairs=True, tuples=False):
    """pairs (the reference's form): the round components declare finalize_logup_in_pairs (round/constraints.rs:116) — 2 x n_logup
    fractions in n_logup columns, degree-3 constraints; the XOR / NOT-AND tables read 3 preprocessed columns and carry 16 multiplicity
    columns paired into 8 logup columns (bitwise_table/constraints.rs:33-71); the rotate table is one fraction over 4 preprocessed
    columns (bit_rotate/mod.rs:77-89, finalize_logup).  pairs=False: round 4's statement, one fraction per logup column everywhere.
    tuples (round 6): the relations at the reference's WIDTHS — every round fraction a 3-wide (XOR / NOT-AND, chips/custom.rs:33-35) or
    4-wide (rotate, :36-37) tuple, the last two the 200-wide state lookups with the numerators (is_padding - 1) and (1 - is_padding)
    (custom.rs:45-46, round/constraints.rs:101-110), the tables' tuples 3 / 4 preprocessed columns (NX_TUPLES_KECCAK)."""
    t = TUPLES_KECCAK if tuples else 0
    if not pairs:
        return [(18 - shift, 8, n_main, 4 * n_logup, 1), (17 - shift, 8, n_main, 4 * n_logup, 1),
                (12 - shift, 2, 6, 8, 1), (12 - shift, 2, 6, 8, 1), (11 - shift, 2, 5, 8, 1)]
    return [(18 - shift, 8, n_main, 4 * n_logup, 1, PAIRS | t), (17 - shift, 8, n_main, 4 * n_logup, 1, PAIRS | t),
            (12 - shift, 3, 16, 32, 1, TABLE | PAIRS | t), (12 - shift, 3, 16, 32, 1, TABLE | PAIRS | t), (11 - shift, 4, 2, 4, 1, TABLE | t)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shift", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--n-main", type=int, default=1000)
    ap.add_argument("--n-logup", type=int, default=500)
    ap.add_argument("--single", action="store_true", help="round 4's statement: one fraction per logup column (finalize_logup) everywhere")
    ap.add_argument("--tuples", action="store_true", help="the relations at the reference's tuple widths: 3 / 4 wide, two 200-wide state lookups per round component (NX_TUPLES_KECCAK)")
    ap.add_argument("--logup-program", action="store_true", help="context option machine.logup_program: every interaction trace from the recorded relation entries")
    ap.add_argument("--check", action="store_true", help="compare the proof with the CPU oracle's, word for word (use with --shift >= 4)")
    a = ap.parse_args()
    import numpy as np
    import nexus_zkvm_amd as nz
    comps = keccak_shaped_components(a.shift, a.n_main, a.n_logup, pairs=not a.single, tuples=a.tuples)
    be = nz.HipBackend(0)
    if a.logup_program:
        be.set_option("machine.logup_program", 1)
    cfg = nz.default_config(pow_bits=10)
    t0 = time.perf_counter(); be.prove_machine(comps, cfg, seed=5); be.sync(); first = time.perf_counter() - t0   # includes the hiprtc compile
    be.sync(); t0 = time.perf_counter()
    for s in range(a.steps):
        words = be.prove_machine(comps, cfg, seed=100 + s)
    be.sync(); el = (time.perf_counter() - t0) / a.steps
    words, st = be.prove_machine(comps, cfg, seed=5, want_stats=True)
    n_cells = sum((c[1] + c[2] + c[3]) << c[0] for c in comps)
    out = {"workload": "keccak-shaped (BASELINE config #5): round components 2^%d / 2^%d rows x (8 + %d + %d columns), tables 2^%d, 2^%d, 2^%d; per-component degree bound 1; %s"
           % (comps[0][0], comps[1][0], a.n_main, 4 * a.n_logup, comps[2][0], comps[3][0], comps[4][0],
              ("one fraction per logup column" if a.single else "logup in PAIRS (finalize_logup_in_pairs: %d fractions per round component, degree-3 constraints), tables over preprocessed columns" % (2 * a.n_logup))
              + ("; tuples 3 / 4 wide and two 200-wide state lookups per round component (NX_TUPLES_KECCAK)" if a.tuples else "; tuples 1 / 2 wide")
              + ("; interaction trace from the recorded relation entries (nx_logup_program)" if a.logup_program else "")),
           "n_columns": sum(c[1] + c[2] + c[3] for c in comps), "trace_cells": n_cells, "ms_per_prove": round(1e3 * el, 3),
           "first_prove_ms_with_jit": round(1e3 * first, 1), "cells_per_s": n_cells / el, "proof_words": int(len(words)),
           "stages_ms": {k: round(st[k], 3) for k in ("trace_gen", "commit", "interaction", "composition", "oods", "quotients", "fri", "pow", "decommit", "total")},
           "lde_kernel_ms": round(st["lde_kernel_ms"], 3), "merkle_kernel_ms": round(st["merkle_kernel_ms"], 3)}
    if a.check:
        import oracle_lib as O, machine_ref
        ref = machine_ref.prove_machine(comps, O.default_cfg(pow_bits=10), seed=5, threads=os.cpu_count() or 4)
        out["proof_equals_oracle"] = bool(len(ref) == len(words) and np.array_equal(ref, words))
    print(json.dumps(out))
    be.close()
    if a.check and not out["proof_equals_oracle"]:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
