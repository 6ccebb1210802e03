"""The reference-shaped synthetic machine of nx_prove_machine (nexus-zkvm_amd/csrc/machine.hip), restated on the CPU oracle.

TEST INFRASTRUCTURE (checker / cpu_baseline only).  An independent statement of the same machine: the AIR is emitted by the
checker's own tests/ref_emitter.py (one instruction per operation into fresh registers — nothing of the product is imported: not
its recorder nexus_zkvm_amd.air_program, not the C++ emitter of machine.hip; only the ABI's opcode numbers are common ground),
the trace and the logup interaction trace come from the oracle (oracle/air.h, oracle/logup.h), the
transcript follows reference prover/src/machine.rs:197-290 through the oracle's prover session (oracle/air_generic.h).  The
proof must equal nx_prove_machine's word for word.

Component (log_size, n_pre, n_main, 4 L[, log constraint-degree bound — 0 / absent = the config's[, logup mode]]).  Fraction f of a
row: den_f = main[a_f] - z (f even) or main[a_f] + alpha main[b_f] - z (f odd); num_f = 1, or -main[m_f] when f % 3 == 2;
a_f = (3 + 7 f) % n_main, b_f = (5 + 11 f) % n_main, m_f = (2 + 13 f) % n_main.  Logup mode (include/nexus_hip.h NX_LOGUP_*):
0 = one fraction per logup column (finalize_logup); PAIRS = two per column (finalize_logup_in_pairs over columns built pairwise like
reference prover2/machine/src/lookups/logup_trace_builder.rs:86-101; | ODD: the last column holds the single left-over fraction);
TABLE = the tuples read PREPROCESSED columns a_f = (2 + 3 f) % n_pre, b_f = (1 + 5 f) % n_pre and every numerator is -main[m_f]
(reference prover/src/extensions/multiplicity.rs:111-124).

Tuple schedule = bits 4..7 of the logup mode (TUPLES(k); include/nexus_hip.h NX_TUPLES_*): the widths and entry kinds of the reference's
own relations.  V1: widths cycling 1, 1, 4, 1, 9, 1, 3, 1 (range256.rs:37; bit_op.rs:31,355: [CONSTANT, b, c, a] with a flag column as
numerator; register_mem_check.rs:34; the 3-wide one = [column, constant 5, column + column]); KECCAK: 3- / 4-wide bitwise lookups
(chips/custom.rs:33-37) and the component's last two fractions 200 wide with the numerators m - 1 and 1 - m (custom.rs:45-46,
extensions/keccak/round/constraints.rs:101-110); V2: prover2's 9 / 21 / 14 / 10 / 4 / 12 / 8 (prover2/machine/src/lookups/
relations.rs:33-90), the 4-wide one = [column, constant 7, column + column, column].  Entry k of fraction f reads column
(3 + 7 f + 5 k) % n_main (a table: (2 + 3 f + 5 k) % n_pre), the second column of a sum (5 + 11 f + 3 k) % n.
"""
import os

import numpy as np

import oracle_lib as O

P = O.P


PAIRS, ODD, TABLE = 1, 2, 4
V1, KECCAK, V2 = 1, 2, 3


def TUPLES(k):
    return k << 4


def logup_mode(comp):
    return comp[5] if len(comp) > 5 else 0


def tuple_sched(comp):
    return (logup_mode(comp) >> 4) & 15


def n_fracs(comp):
    L, mode = comp[3] // 4, logup_mode(comp)
    if not (mode & PAIRS) or L == 0:
        return L
    return 2 * L - (1 if mode & ODD else 0)


# numerator kinds: the fraction's numerator as (sign, uses the multiplicity column, constant term): sign * m + constant
ONE, NEG_M, M_MINUS_1, ONE_MINUS_M, PLUS_M = (0, 1), (-1, 0), (1, -1), (-1, 1), (1, 0)     # (coefficient of m, constant)


def frac_shape(comp, f):
    """(tree of the tuple columns, entries, numerator (coefficient of main[m], constant), m) of fraction f.
    An entry is ('col', k), ('const', c) or ('sum', k, k2)."""
    n_pre, n_main, mode, sched = comp[1], comp[2], logup_mode(comp), tuple_sched(comp)
    table = bool(mode & TABLE)
    m = (2 + 13 * f) % n_main
    if sched == 0:
        if table:
            tup = [(2 + 3 * f) % n_pre] + ([(1 + 5 * f) % n_pre] if f & 1 else [])
            return 0, [("col", k) for k in tup], NEG_M, m
        tup = [(3 + 7 * f) % n_main] + ([(5 + 11 * f) % n_main] if f & 1 else [])
        return 1, [("col", k) for k in tup], (NEG_M if f % 3 == 2 else ONE), m
    F = n_fracs(comp)
    n = n_pre if table else n_main
    state = sched == KECCAK and not table and F >= 2 and f >= F - 2
    if sched == V1:
        w = (1, 1, 4, 1, 9, 1, 3, 1)[f % 8]
    elif sched == V2:
        w = (9, 21, 14, 10, 4, 12, 8)[f % 7]
    else:
        w = 200 if state else (4 if f % 4 == 3 else 3)
    first = (2 + 3 * f) if table else (3 + 7 * f)
    ent = [("col", (first + 5 * k) % n) for k in range(w)]

    def as_sum(k):
        ent[k] = ("sum", ent[k][1], (5 + 11 * f + 3 * k) % n)

    if sched == V1 and w == 4:
        ent[0] = ("const", 1 + f % 3)
    if sched == V1 and w == 3:
        ent[1] = ("const", 5); as_sum(2)
    if sched == V2 and w == 4:
        ent[1] = ("const", 7); as_sum(2)
    num = NEG_M if (table or f % 3 == 2) else ONE
    if not table:
        if sched == V1 and f % 8 == 2:
            num = PLUS_M
        if sched == V2 and f % 5 == 1 and f % 3 != 2:
            num = PLUS_M
        if state:
            num = M_MINUS_1 if f == F - 2 else ONE_MINUS_M
    return (0 if table else 1), ent, num, m


def frac_def(comp, f):
    """(tree of the tuple columns, tuple column indices, multiplicity main column or None) of fraction f — schedule 0's form"""
    tree, ent, num, m = frac_shape(comp, f)
    assert all(e[0] == "col" for e in ent)
    return tree, [e[1] for e in ent], (m if num != ONE else None)


def batches(comp):
    """the fractions of every logup column (finalize_logup_batched's batching: one per column, or in pairs)"""
    F, L = n_fracs(comp), comp[3] // 4
    if logup_mode(comp) & PAIRS:
        return [list(range(2 * j, min(F, 2 * j + 2))) for j in range(L)]
    return [[j] for j in range(L)]


def machine_component(ap, comp, loc, z, alpha, shift, cfg_lcd=1):
    """The component's AIR through the recording evaluator: what `add_constraints` of a FrameworkEval would declare.
    A component whose degree bound (its own, or the config's `cfg_lcd` when 0 / absent) is 2 has degree-4 constraints: each degree-2
    main-trace constraint times the two columns it squares, each transition constraint times main0 main1 (degree 4 over a neighbour
    row); the logup constraints stay degree 2 (finalize_logup) or 3 (in pairs)."""
    log, n_pre, n_main, n_inter = comp[:4]
    bound = comp[4] if len(comp) > 4 and comp[4] else cfg_lcd
    quartic = bound >= 2
    L = n_inter // 4
    pre0, main0, inter0 = loc
    pb = ap.Emitter()
    PRE, MAIN, INT = 0, n_pre, n_pre + n_main
    m0, m0n = pb.next_trace_mask(MAIN + 0, (0, 1))
    m1, m1n = pb.next_trace_mask(MAIN + 1, (0, 1))
    (is_last,) = pb.next_trace_mask(PRE + 1)
    not_last = 1 - is_last
    t0, t1 = (m0n - m0 - 1) * not_last, (m1n - m1 - m0) * not_last
    pb.add_constraint(t0 * m0 * m1 if quartic else t0)          # +2 components: degree 4 over a neighbour row (the reference's Pc / IsPadding constraints)
    pb.add_constraint(t1 * m0 * m1 if quartic else t1)
    main = [m0, m1] + [pb.next_trace_mask(MAIN + k)[0] for k in range(2, n_main)]
    for k in range(2, n_main):
        if k % 16 >= 2:
            c2 = main[k] - main[k - 1] * main[k - 1] - main[k - 2] * main[k - 2]
            pb.add_constraint(c2 * main[k - 1] * main[k - 2] if quartic else c2)
    if L:
        ze, al, sh = pb.econst(z), pb.econst(alpha), pb.econst(shift)
        pre = {}                                                   # preprocessed columns a table component's tuples read (offset 0)

        def tuple_col(tree, k):
            if tree == 1:
                return main[k]
            if k not in pre:
                pre[k] = pb.next_trace_mask(PRE + k)[0]
            return pre[k]

        apow = {}                                                  # LookupElements' alpha powers are constants of the relation

        def alpha_pow(k):
            if k == 1:
                return al
            if k not in apow:
                v = np.array([1, 0, 0, 0], np.uint32)
                for _ in range(k):
                    v = O.qm31_mul(v, alpha)
                apow[k] = pb.econst(v)
            return apow[k]

        def entry(tree, en):
            if en[0] == "col":
                return tuple_col(tree, en[1])
            if en[0] == "const":
                return pb.const(en[1])
            return tuple_col(tree, en[1]) + tuple_col(tree, en[2])

        prev = None
        for j, fs in enumerate(batches(comp)):
            # Fraction sum of the batch (stwo-constraint-framework Fraction::add): (n0 d1 + n1 d0) / (d0 d1); a single fraction as it is
            num_neg, den = None, None                              # - numerator, denominator
            for f in fs:
                tree, ent, (cm, c0), m = frac_shape(comp, f)
                if tuple_sched(comp) == 0:
                    tup = [e[1] for e in ent]
                    d = tuple_col(tree, tup[0]) - ze if len(tup) == 1 else al * tuple_col(tree, tup[1]) + tuple_col(tree, tup[0]) - ze   # E arithmetic: B - E lowers to (-E) + B
                    nn = main[m] if cm else pb.const(P - 1)                                   # - num
                else:
                    # Relation::combine: sum_k alpha^k value_k - z, the values in declaration order
                    d = None
                    for k, en in enumerate(ent):
                        v = entry(tree, en)
                        term = v if k == 0 else alpha_pow(k) * v
                        d = term if d is None else d + term
                    d = d - ze
                    # - numerator, the numerator being cm * main[m] + c0
                    nn = {ONE: lambda: pb.const(P - 1), NEG_M: lambda: main[m], PLUS_M: lambda: 0 - main[m],
                          M_MINUS_1: lambda: 1 - main[m], ONE_MINUS_M: lambda: main[m] - 1}[(cm, c0)]()
                if den is None:
                    num_neg, den = nn, d
                else:
                    num_neg, den = num_neg * d + nn * den, den * d
            if j + 1 < L:
                (cur,) = pb.next_secure_mask(INT + 4 * j)
                diff = cur if prev is None else cur - prev
            else:
                prow, cur = pb.next_secure_mask(INT + 4 * j, (-1, 0))
                diff = cur - prow
                if prev is not None:
                    diff = diff - prev
                diff = diff + sh
            pb.add_constraint(diff * den + num_neg)
            prev = cur
    cols = [(0, pre0 + k) for k in range(n_pre)] + [(1, main0 + k) for k in range(n_main)] + [(2, inter0 + k) for k in range(n_inter)]
    prog = pb.build()
    # mask lists exactly as the machine declares them (columns the program happens not to load are still sampled at offset 0)
    masks = [[0]] * n_pre + [[0, 1], [0, 1]] + [[0]] * (n_main - 2) + [([-1, 0] if k // 4 + 1 == L else [0]) for k in range(n_inter)]
    return ap.Component(log, prog, cols, masks, log_constraint_degree_bound=comp[4] if len(comp) > 4 else 0)


def interaction_trace(comp, main_cols, z, alpha, pre_cols=None):
    """LogupTraceGenerator as the reference drives it (one fraction per column or a merged pair, finalize_col, finalize_last) on the
    oracle.  main_cols / pre_cols: the component's finalized main-trace / preprocessed columns.  Returns (4 L coordinate columns,
    claimed sum)."""
    log, n_pre, n_main, n_inter = comp[:4]
    L = n_inter // 4
    if L == 0:
        return [], np.zeros(4, np.uint32)
    wmax = max([2] + [len(frac_shape(comp, f)[1]) for f in range(n_fracs(comp))])
    pw = [np.array([1, 0, 0, 0], np.uint32)]
    for _ in range(1, wmax):
        pw.append(O.qm31_mul(pw[-1], alpha))
    ap = np.array(pw, np.uint32)
    cols, prev = [], None
    for fs in batches(comp):
        args = []
        for f in fs:
            tree, ent, (cm, c0), m = frac_shape(comp, f)
            src = main_cols if tree == 1 else pre_cols
            n_rows = len(main_cols[0])
            vals = []                                                     # the tuple's VALUES, entry by entry (what the chip's generator computes)
            for en in ent:
                if en[0] == "col":
                    vals.append(src[en[1]])
                elif en[0] == "const":
                    vals.append(np.full(n_rows, en[1], np.uint32))
                else:
                    vals.append(((src[en[1]].astype(np.uint64) + src[en[2]]) % P).astype(np.uint32))
            den = O.logup_combine(vals, ap[:len(vals)], z)
            if cm == 0:
                args.append((den, (c0 % P, 0, 0, 0), None))
            elif c0 == 0:
                args.append((den, (cm % P, 0, 0, 0), main_cols[m]))
            else:                                                         # m - 1 / 1 - m: the numerator column itself
                numer = ((cm * main_cols[m].astype(np.int64) + c0) % P).astype(np.uint32)
                args.append((den, (1, 0, 0, 0), numer))
        if len(args) == 2:                                                # LogupTraceBuilder: (a d + b c) / (b d)
            (da, sa, ma), (db, sb, mb) = args
            col = O.logup_finalize_col(da, scale_a=sa, mult_a=ma, den_b=db, scale_b=sb, mult_b=mb, prev=prev)
        else:
            (da, sa, ma), = args
            col = O.logup_finalize_col(da, scale_a=sa, mult_a=ma, prev=prev)
        cols.append(col); prev = col
    cols[-1], claimed = O.logup_finalize_last(cols[-1])
    return [c for col in cols for c in col], claimed


def prove_machine(comps, cfg, seed=1, ad=b"", threads=None, component_fn=None):
    """nexus_vm_prover::prove for the machine, on the CPU oracle: returns the NXP1 proof words.
    component_fn(ap, comp, loc, z, alpha, shift, cfg_lcd) (default: machine_component, the checker's own emission) builds a component."""
    import ref_emitter as ap            # the checker's own emitter: the product's recorder is not imported
    threads = threads or max(4, os.cpu_count() or 4)
    O.lib().orc_logup_set_threads(threads)
    comps = [tuple(int(x) for x in c) for c in comps]
    s = O.ProverSession(cfg, max(c[0] for c in comps), threads)
    for byte in ad:
        s.mix_u64(byte)                                       # machine.rs:198-200
    for c in comps:
        s.mix_u64(c[0])                                       # machine.rs:204-206
    pre = O.synth_tree_columns(comps, 0, seed, threads=threads)
    s.commit(pre)                                             # :208-228
    main = O.synth_tree_columns(comps, 1, seed, threads=threads)
    s.commit(main)                                            # :230-237 (the session copies: `main` keeps the evaluations, like the reference's clone)
    z, alpha = s.draw_felts(2)                                # :239-240
    inter, claimed, shifts, off, poff = [], [], [], 0, 0
    for c in comps:
        cols, cs = interaction_trace(c, main[off:off + c[2]], z, alpha, pre[poff:poff + c[1]])
        off += c[2]; poff += c[1]
        inter += cols; claimed.append(cs)
        n_inv = pow((1 << c[0]) % P, P - 2, P)
        shifts.append(np.array([(int(x) * n_inv) % P for x in cs], np.uint32))
    s.mix_felts(np.array(claimed, np.uint32))                 # :262
    s.commit(inter)                                           # :263
    locs, a, b, d = [], 0, 0, 0
    for c in comps:
        locs.append((a, b, d)); a += c[1]; b += c[2]; d += c[3]
    components = [(component_fn or machine_component)(ap, c, l, z, alpha, sh, int(cfg[6])) for c, l, sh in zip(comps, locs, shifts)]
    return s.prove(components)                                # :286-290


def verify_machine(comps, cfg, words, claimed, ad=b""):
    """`nexus_vm_prover::verify` for the machine (reference prover/src/machine.rs:363-500) on the oracle's VERIFIER session: the
    transcript prefix replayed from the proof's own commitments (:437-482: ad bytes, log sizes, preprocessed and main roots, lookup
    elements, claimed sums, interaction root), the components rebuilt from the drawn lookup elements and the claimed sums
    (`claimed`: (n_components, 4) words — `Proof.claimed_sum`, nx_machine_claimed_sums), then core::verifier::verify.  Costs KBs of
    hashing whatever the trace size: the size-independent check of a proof the oracle PROVER could not reproduce in test time.
    None when accepted, else the verifier's error text."""
    import ref_emitter as ap
    comps = [tuple(int(x) for x in c) for c in comps]
    words = np.ascontiguousarray(words, dtype=np.uint32)
    claimed = np.ascontiguousarray(claimed, dtype=np.uint32).reshape(len(comps), 4)
    h = O.proof_header_words()
    if int(words[h]) != 4:
        return "proof does not hold 4 commitments"
    roots = [words[h + 1 + 8 * t:h + 9 + 8 * t] for t in range(3)]
    v = O.VerifierSession(cfg)
    for byte in ad:
        v.mix_u64(byte)
    for c in comps:
        v.mix_u64(c[0])
    tree_logs = [[c[0] for c in comps for _ in range(c[1 + t])] for t in range(3)]
    v.commit(roots[0], tree_logs[0])
    v.commit(roots[1], tree_logs[1])
    z, alpha = v.draw_felts(2)
    v.mix_felts(claimed)
    v.commit(roots[2], tree_logs[2])
    locs, a, b, d = [], 0, 0, 0
    for c in comps:
        locs.append((a, b, d)); a += c[1]; b += c[2]; d += c[3]
    shifts = []
    for c, cs in zip(comps, claimed):
        n_inv = pow((1 << c[0]) % P, P - 2, P)
        shifts.append(np.array([(int(x) * n_inv) % P for x in cs], np.uint32))
    components = [machine_component(ap, c, l, z, alpha, sh, int(cfg[6])) for c, l, sh in zip(comps, locs, shifts)]
    return v.verify(components, words)
