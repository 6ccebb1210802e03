"""AIRs recorded through nexus_zkvm_amd.air_program, with valid traces, for the prover-session tests (CPU oracle and GPU).

synthetic_component: the synthetic machine of oracle/air.h as a generic Component (its proof must equal prove_synth's).
logup_component:     a small lookup-style component: main columns a, b, c with c = a*b + 3, a free multiplicity-less logup
                     column S (secure, 4 coordinates in the interaction tree) with the running-sum constraint
                     (S(row) - S(row-1) + shift) * (z - a - alpha*b) = 1, S in natural trace order minus (row+1)*shift so that it
                     wraps to zero — the shape of stwo-constraint-framework's logup constraints (mask [-1, 0] on the
                     interaction column, lookup elements z/alpha and claimed-sum shift as secure constants)."""
import numpy as np

import oracle_lib as O

P = O.P


def qmul(a, b):
    out = np.zeros(4, np.uint32)
    O.lib().orc_qm31_mul(O.ptr(O.u32(a)), O.ptr(O.u32(b)), O.ptr(out))
    return out


def qinv(a):
    out = np.zeros(4, np.uint32)
    O.lib().orc_qm31_inv(O.ptr(O.u32(a)), O.ptr(out))
    return out


def qadd(a, b):
    return ((np.asarray(a, np.uint64) + np.asarray(b, np.uint64)) % P).astype(np.uint32)


def qsub(a, b):
    return ((np.asarray(a, np.uint64) + P - np.asarray(b, np.uint64)) % P).astype(np.uint32)


def synthetic_component(ap, log, n_pre, n_main, n_inter, pre0=0, main0=0, inter0=0):
    from test_air_program_cpu import synthetic_program
    prog = synthetic_program(ap, n_pre, n_main, n_inter)
    cols = [(0, pre0 + k) for k in range(n_pre)] + [(1, main0 + k) for k in range(n_main)] + [(2, inter0 + k) for k in range(n_inter)]
    return ap.Component(log, prog, cols)


def logup_main_trace(log, seed):
    """a, b random, c = a*b + 3 — natural order, finalized to bit-reversed circle-domain order."""
    rng = np.random.default_rng(seed)
    n = 1 << log
    a = rng.integers(0, P, n, dtype=np.uint64)
    b = rng.integers(0, P, n, dtype=np.uint64)
    c = (a * b + 3) % P
    nat = [a.astype(np.uint32), b.astype(np.uint32), c.astype(np.uint32)]
    return nat, [O.finalize_column(x) for x in nat]


def logup_interaction_trace(log, nat_main, z, alpha):
    """Returns (4 coordinate columns of S finalized, shift)."""
    n = 1 << log
    a, b = nat_main[0], nat_main[1]
    fr = []
    for i in range(n):
        den = qsub(qsub(z, [a[i], 0, 0, 0]), qmul(alpha, [b[i], 0, 0, 0]))
        fr.append(qinv(den))
    total = np.zeros(4, np.uint32)
    for f in fr:
        total = qadd(total, f)
    shift = qmul(total, [pow(n, P - 2, P), 0, 0, 0])
    S = np.zeros((n, 4), np.uint32)
    run = np.zeros(4, np.uint32)
    for i in range(n):
        run = qsub(qadd(run, fr[i]), shift)
        S[i] = run
    assert not S[n - 1].any()
    return [O.finalize_column(S[:, k].copy()) for k in range(4)], shift


def logup_component(ap, log, z, alpha, shift, main0=0, inter0=0, high_degree=False):
    """high_degree: the same relation also as a degree-3 and a degree-4 constraint, interleaved with the others (what a component with a
    constraint-degree bound of 2 is for): a degree-aware prover evaluates the parts on different domains, the alpha powers stay put."""
    pb = ap.ProgramBuilder()
    (a,) = pb.next_trace_mask(0)
    (b,) = pb.next_trace_mask(1)
    (c,) = pb.next_trace_mask(2)
    s_prev, s_cur = pb.next_secure_mask(3, (-1, 0))
    if high_degree:
        pb.add_constraint((c - a * b - 3) * a * b)
    pb.add_constraint(c - a * b - 3)
    if high_degree:
        pb.add_constraint((c - a * b - 3) * c)
    ze, al, sh = pb.econst(z), pb.econst(alpha), pb.econst(shift)
    den = ze - a - al * b
    pb.add_constraint((s_cur - s_prev + sh) * den - 1)
    if high_degree:
        pb.add_constraint(((s_cur - s_prev + sh) * den - 1) * a * c)
    cols = [(1, main0), (1, main0 + 1), (1, main0 + 2)] + [(2, inter0 + k) for k in range(4)]
    return ap.Component(log, pb.build(), cols)


def tree_count_statement(ap, n_trees, log=6, seed=11):
    """The synthetic machine's AIR over 2 or 4 trace trees instead of the reference's 3 (Stwo takes any TreeVec; the session takes the
    count it is given): n_trees == 2: preprocessed + main only (no interaction columns); n_trees == 4: the last six main columns live
    in a FOURTH tree committed after the interaction tree.  Returns (trees: list of column lists in commit order, component)."""
    n_pre, n_main, n_inter = 3, 20, (0 if n_trees == 2 else 8)
    comps = [(log, n_pre, n_main, n_inter)]
    pre, main = O.synth_tree_columns(comps, 0, seed), O.synth_tree_columns(comps, 1, seed)
    comp = synthetic_component(ap, log, n_pre, n_main, n_inter)
    if n_trees == 2:
        return [pre, main], comp
    inter = O.synth_tree_columns(comps, 2, seed, 5)
    if n_trees == 3:
        return [pre, main, inter], comp
    split = n_main - 6
    cols = [(3, i - split) if (t == 1 and i >= split) else (t, i) for t, i in comp.cols]
    return [pre, main[:split], inter, main[split:]], ap.Component(log, comp.program, cols, comp.masks)


# ---- a lookup-heavy component declared through the recorder's relation API (the reference's way: add_to_relation + finalize_logup*) ----
RELATION_COLS = 7        # main columns a, b, c, d, m, p, q


def relation_main_trace(log, seed):
    """a, b, d, m free; c = a*b + 3; p = a + d (so that a constraint ties it); q free.  Natural order + finalized."""
    rng = np.random.default_rng(seed)
    n = 1 << log
    a, b, d, q = (rng.integers(0, P, n, dtype=np.uint64) for _ in range(4))
    m = rng.integers(0, 1 << 12, n, dtype=np.uint64)
    c = (a * b + 3) % P
    p = (a + d) % P
    nat = [x.astype(np.uint32) for x in (a, b, c, d, m, p, q)]
    return nat, [O.finalize_column(x) for x in nat]


def relation_program(ap, z, alpha, shift, batching="pairs", main0=0, inter0=RELATION_COLS):
    """The recorder run over a component with 5 relation entries of 2 relations: tuples of 1, 2 and 3 values built from EXPRESSIONS
    (a + 5, a value at the NEXT row), multiplicities 1, -m and the expression (q - 1) — what the reference's chips declare
    (prover/src/extensions/keccak/round/constraints.rs:95-116: (is_padding - 1); bitwise_table/constraints.rs:50-71: al + const).
    Returns the ProgramBuilder (build() = constraints, build_logup() = the fraction program)."""
    pb = ap.ProgramBuilder()
    a, a_next = pb.next_trace_mask(main0 + 0, (0, 1))
    (b,) = pb.next_trace_mask(main0 + 1)
    (c,) = pb.next_trace_mask(main0 + 2)
    (d,) = pb.next_trace_mask(main0 + 3)
    (m,) = pb.next_trace_mask(main0 + 4)
    (p,) = pb.next_trace_mask(main0 + 5)
    (q,) = pb.next_trace_mask(main0 + 6)
    pb.add_constraint(c - a * b - 3)
    pb.add_constraint(p - a - d)
    r3 = pb.relation(z, alpha, 3)
    r2 = pb.relation([int(x) ^ 1 for x in z], alpha, 2)         # a second relation: its own z
    pb.add_to_relation(r3, 1, [a, b, c])
    pb.add_to_relation(r3, -m, [a + 5, d, a_next])
    pb.add_to_relation(r2, q - 1, [p, b])
    pb.add_to_relation(r2, 1, [d])
    pb.add_to_relation(r3, -m, [b, 7])
    if batching == "pairs":
        pb.finalize_logup_in_pairs(inter0, shift)
    elif batching == "single":
        pb.finalize_logup(inter0, shift)
    else:
        pb.finalize_logup_batched(inter0, shift, batching)
    return pb


def relation_component(ap, log, pb, n_logup_cols):
    cols = [(1, k) for k in range(RELATION_COLS)] + [(2, k) for k in range(4 * n_logup_cols)]
    return ap.Component(log, pb.build(), cols)
