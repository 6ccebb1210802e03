"""Logup interaction-trace restatement (oracle/logup.h) pinned by the identities the construction
rests on (no Stwo here to compare with): the fractions really are num/den, merged pairs are sums of fractions, the running
column is a sum over columns, the claimed sum is the sum of all fractions, and after finalize_last the column is a prefix
sum in natural coset order whose last row is zero (sum of (x - mean) over all rows)."""
import os

import numpy as np
import pytest

import oracle_lib as O

P = O.P




def q_mul_py(x, y):
    """QM31 = CM31[u]/(u^2 - 2 - i), CM31 = M31[i]/(i^2 + 1); plain Python ints."""
    def cmul(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
    def cadd(a, b):
        return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
    xa, xb, ya, yb = (x[0], x[1]), (x[2], x[3]), (y[0], y[1]), (y[2], y[3])
    bb = cmul(xb, yb)
    r = ((2 * bb[0] - bb[1]) % P, (2 * bb[1] + bb[0]) % P)       # * (2 + i)
    a = cadd(cmul(xa, ya), r)
    b = cadd(cmul(xa, yb), cmul(xb, ya))
    return (a[0], a[1], b[0], b[1])


def q_add_py(x, y):
    return tuple((int(a) + int(b)) % P for a, b in zip(x, y))


def rows(col4):
    return [tuple(int(col4[q][r]) for q in range(4)) for r in range(len(col4[0]))]


def coset_row_position(c, log):
    n = 1 << log
    d = c // 2 if c % 2 == 0 else n - 1 - (c - 1) // 2
    return int(format(d, "0%db" % log)[::-1], 2) if log else 0


@pytest.mark.parametrize("log", [4, 7])
def test_logup_identities(oracle, log):
    rng = np.random.default_rng(log)
    n = 1 << log
    tuple_cols = rng.integers(0, P, (3, n), dtype=np.uint32)
    alphas = rng.integers(0, P, (3, 4), dtype=np.uint32)
    z = rng.integers(0, P, 4, dtype=np.uint32)
    den = oracle.logup_combine(list(tuple_cols), alphas, z)
    for r in (0, 1, n - 1):
        acc = (0, 0, 0, 0)
        for k in range(3):
            acc = q_add_py(acc, q_mul_py(tuple(int(x) for x in alphas[k]), (int(tuple_cols[k][r]), 0, 0, 0)))
        assert rows(den)[r] == tuple((a - int(b)) % P for a, b in zip(acc, z))
    mult = rng.integers(0, 5, n, dtype=np.uint32)
    # one fraction: col * den == num
    col0 = oracle.logup_finalize_col(den, scale_a=(P - 1, 0, 0, 0), mult_a=mult)
    for r in (0, 3, n - 1):
        assert q_mul_py(rows(col0)[r], rows(den)[r]) == ((P - int(mult[r])) % P, 0, 0, 0)
    # two merged fractions + previous column: (col1 - col0) * denA * denB == numA * denB + numB * denA
    den_b = oracle.logup_combine(list(tuple_cols[:2]), alphas[:2], z)
    col1 = oracle.logup_finalize_col(den, den_b=den_b, mult_b=mult, prev=col0)
    for r in (0, 5, n - 1):
        diff = tuple((a - b) % P for a, b in zip(rows(col1)[r], rows(col0)[r]))
        lhs = q_mul_py(q_mul_py(diff, rows(den)[r]), rows(den_b)[r])
        rhs = q_add_py(rows(den_b)[r], q_mul_py((int(mult[r]), 0, 0, 0), rows(den)[r]))
        assert lhs == rhs
    # finalize_last: claimed sum = sum of the column; result = prefix sums of (x - mean) along natural coset order
    last, claimed = oracle.logup_finalize_last(col1)
    total = (0, 0, 0, 0)
    for v in rows(col1):
        total = q_add_py(total, v)
    assert tuple(int(x) for x in claimed) == total
    inv_n = pow(n, P - 2, P)
    mean = tuple(t * inv_n % P for t in total)
    run = (0, 0, 0, 0)
    lr, c1 = rows(last), rows(col1)
    for c in range(n):
        p = coset_row_position(c, log)
        run = q_add_py(run, tuple((a - b) % P for a, b in zip(c1[p], mean)))
        assert lr[p] == run
    assert lr[coset_row_position(n - 1, log)] == (0, 0, 0, 0)


# ---------------- the interaction trace FROM THE RECORDED AIR (nx_logup_program / oracle logup_program) ----------------

def _relation_statement(log, batching, seed=77):
    """Drive the oracle session: main tree, lookup elements, interaction trace from the fraction program, prove.  Returns (proof, pb)."""
    import nexus_zkvm_amd.air_program as ap
    import air_examples as AE
    # a batch of three fractions makes a degree-4 constraint (diff d0 d1 d2): the bound +2; pairs are degree 3 under +1 like the reference's
    lcd = 1 if isinstance(batching, str) else 2
    cfg = O.default_cfg(pow_bits=3, log_constraint_degree=lcd)
    nat, fin = AE.relation_main_trace(log, seed)
    s = O.ProverSession(cfg, log, 2)
    s.mix_u64(log)
    s.commit([])
    s.commit(fin)
    z, alpha = s.draw_felts(2)
    # the fraction program does not depend on the claimed sum: record once with a zero shift, generate the trace, then record the
    # constraints with the real shift (the reference builds its components after the interaction trace too, machine.rs:264-285)
    pb0 = AE.relation_program(ap, z, alpha, (0, 0, 0, 0), batching)
    frac = pb0.build_logup()
    cols = O.logup_program(frac, fin + [None] * (4 * frac.n_logup_cols), log, frac.n_logup_cols)
    cols[-1], claimed = O.logup_finalize_last(cols[-1])
    n_inv = pow((1 << log) % P, P - 2, P)
    shift = [(int(x) * n_inv) % P for x in claimed]
    s.mix_felts(np.array([claimed], np.uint32))
    s.commit([c for col in cols for c in col])
    pb = AE.relation_program(ap, z, alpha, shift, batching)
    return s.prove([AE.relation_component(ap, log, pb, frac.n_logup_cols)]), frac, (z, alpha), fin


V2_MAIN_POS = {0: 0, 2: 1, 3: 2, 4: 3, 5: 4, 6: 5}          # component-local column -> place among the component's six main-tree columns


def _v2_trees(logs, seed):
    """A prover2-shaped statement (reference prover2/machine/src/prove.rs:70-84): every component brings its own log size, ONE column of
    the preprocessed tree (the relation example's column b) and six of the main tree; trees are the components' columns, component after
    component.  Returns (finalized columns per component, tree 0, tree 1)."""
    import air_examples as AE
    fins = [AE.relation_main_trace(log, seed + c)[1] for c, log in enumerate(logs)]
    return fins, [f[1] for f in fins], [f[k] for f in fins for k in sorted(V2_MAIN_POS)]


def _v2_component_cols(c, n_logup_cols, inter_base):
    return [(0, c) if k == 1 else (1, 6 * c + V2_MAIN_POS[k]) for k in range(7)] + [(2, inter_base + j) for j in range(4 * n_logup_cols)]


def _v2_statement(logs, batchings, seed=31):
    """prove.rs:34-135 on the oracle session with the interaction trace generated from the recorded relation entries, per component
    (what rust/nexus-hip/reference_patch/prove2_hip.rs does on the device).  Returns (proof, cfg, roots, tree log sizes, components)."""
    import nexus_zkvm_amd.air_program as ap
    import air_examples as AE
    cfg = O.default_cfg(pow_bits=3, log_constraint_degree=1)
    fins, tree0, tree1 = _v2_trees(logs, seed)
    s = O.ProverSession(cfg, max(logs), 2)
    for log in logs:
        s.mix_u64(log)
    roots = [s.commit(tree0), s.commit(tree1)]
    z, alpha = s.draw_felts(2)
    tree2, claimed_all, shifts, n_cols = [], [], [], []
    for log, fin, batching in zip(logs, fins, batchings):
        frac = AE.relation_program(ap, z, alpha, (0, 0, 0, 0), batching).build_logup()
        cols = O.logup_program(frac, fin + [None] * (4 * frac.n_logup_cols), log, frac.n_logup_cols)
        cols[-1], claimed = O.logup_finalize_last(cols[-1])
        n_inv = pow((1 << log) % P, P - 2, P)
        shifts.append([(int(x) * n_inv) % P for x in claimed])
        claimed_all.append(claimed); n_cols.append(frac.n_logup_cols)
        tree2 += [x for col in cols for x in col]
    s.mix_felts(np.array(claimed_all, np.uint32))
    roots.append(s.commit(tree2))
    comps, base = [], 0
    for c, (log, batching) in enumerate(zip(logs, batchings)):
        comps.append(ap.Component(log, AE.relation_program(ap, z, alpha, shifts[c], batching).build(), _v2_component_cols(c, n_cols[c], base)))
        base += 4 * n_cols[c]
    tree_logs = [list(logs), [log for log in logs for _ in range(6)], [log for log, n in zip(logs, n_cols) for _ in range(4 * n)]]
    return s.prove(comps), cfg, roots, tree_logs, comps, np.array(claimed_all, np.uint32)


def test_prover2_shaped_statement_with_generated_interaction_traces_verifies(oracle):
    """Three components of their own sizes, each with a preprocessed column, pairs and single-fraction columns side by side, every
    interaction trace generated from the component's own recorded relation entries: the oracle prover accepts the traces (its
    constraint check) and the oracle VERIFIER accepts the proof from the roots and the claimed sums alone."""
    logs, batchings = (7, 9, 8), ("pairs", "single", "pairs")
    proof, cfg, roots, tree_logs, comps, claimed = _v2_statement(logs, batchings)
    assert _v2_verify(cfg, logs, roots, tree_logs, claimed, comps, proof) is None
    bad = proof.copy(); bad[len(bad) // 2] ^= 1
    assert _v2_verify(cfg, logs, roots, tree_logs, claimed, comps, bad) is not None
    wrong = claimed.copy(); wrong[1, 0] ^= 1                     # a claimed sum the transcript does not hold
    assert _v2_verify(cfg, logs, roots, tree_logs, wrong, comps, proof) is not None


def _v2_verify(cfg, logs, roots, tree_logs, claimed, comps, words):
    v = O.VerifierSession(cfg)
    for log in logs:
        v.mix_u64(log)
    v.commit(roots[0], tree_logs[0]); v.commit(roots[1], tree_logs[1])
    v.draw_felt(); v.draw_felt()
    v.mix_felts(claimed)
    v.commit(roots[2], tree_logs[2])
    return v.verify(comps, words)


@pytest.mark.parametrize("batching", ["pairs", "single", [0, 0, 0, 1, 2], [2, 0, 1, 1, 0]])
def test_interaction_trace_from_the_recorded_relation_entries_satisfies_its_constraints(oracle, batching):
    """the reference fills its interaction trace with hand-written per-chip generators that mirror the relation entries
    of the AIR.  Here the entries themselves — recorded by the same evaluator that records the constraints — are lowered to a fraction
    program and THAT is the generator: expression multiplicities ((q - 1), -m), tuple values that are expressions or next-row reads,
    two relations, any batching.  The oracle prover's constraint check (ProvingError::ConstraintsNotSatisfied) accepts the trace."""
    proof, frac, _, _ = _relation_statement(6, batching)
    assert len(proof) > 100
    n_cols = {"pairs": 3, "single": 5}.get(batching if isinstance(batching, str) else "", 3)
    assert frac.n_logup_cols == n_cols


def test_fraction_program_equals_the_hand_written_generator(oracle):
    """The same columns as LogupTraceGenerator driven by hand (combine + finalize_col, tests/oracle_lib.py): one fraction per column,
    tuples of raw columns, numerators 1 and -m."""
    import nexus_zkvm_amd.air_program as ap
    rng = np.random.default_rng(9)
    log, n = 7, 128
    cols = [rng.integers(0, P, n, dtype=np.uint32) for _ in range(4)]
    z, alpha = rng.integers(0, P, 4, dtype=np.uint32), rng.integers(0, P, 4, dtype=np.uint32)
    pb = ap.ProgramBuilder()
    v = [pb.next_trace_mask(k)[0] for k in range(4)]
    rel = pb.relation(z, alpha, 3)
    pb.add_to_relation(rel, 1, [v[0], v[1], v[2]])
    pb.add_to_relation(rel, -v[3], [v[1]])
    pb.finalize_logup(4, (0, 0, 0, 0))
    frac = pb.build_logup()
    got = O.logup_program(frac, cols + [None] * 8, log, 2)
    apw = np.stack([np.array([1, 0, 0, 0], np.uint32), alpha, O.qm31_mul(alpha, alpha)])
    c0 = O.logup_finalize_col(O.logup_combine(cols[:3], apw, z))
    c1 = O.logup_finalize_col(O.logup_combine([cols[1]], apw[:1], z), scale_a=(P - 1, 0, 0, 0), mult_a=cols[3], prev=c0)
    assert all(np.array_equal(a, b) for a, b in zip(got[0], c0)) and all(np.array_equal(a, b) for a, b in zip(got[1], c1))


def test_generated_logup_kernel_source_compiles_for_gfx950(oracle, tmp_path):
    """nx_logup_program's generated HIP (no GPU needed for the text) through an offline hipcc for gfx950; a small "air.segment" budget
    forces several kernels: each later one starts from the running sum its predecessor stored.  Malformed programs are refused."""
    import shutil, subprocess
    import nexus_zkvm_amd as nz
    import nexus_zkvm_amd.air_program as ap
    import air_examples as AE
    pb = AE.relation_program(ap, (1, 2, 3, 4), (5, 6, 7, 8), (0, 0, 0, 0), "pairs")
    frac = pb.build_logup()
    src = nz.logup_program_source(frac, AE.RELATION_COLS + 12, 3)
    assert src.count("__attribute__((global))") == 1 and "trace_row_offset" in src and "m_inv(" in src
    f = tmp_path / "logup.hip"
    f.write_text(src)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "--offload-arch=gfx950", "--cuda-device-only", "-O3", "-c", str(f), "-o", str(tmp_path / "l.o")], check=True, timeout=300)
    os.environ["NX_AIR_SEGMENT"] = "200"
    try:
        many = nz.logup_program_source(frac, AE.RELATION_COLS + 12, 3)
    finally:
        del os.environ["NX_AIR_SEGMENT"]
    assert many.count("__attribute__((global))") == 3 and "Q run = {out[" in many
    (tmp_path / "logup3.hip").write_text(many)
    subprocess.run([hipcc, "--offload-arch=gfx950", "--cuda-device-only", "-O3", "-c", str(tmp_path / "logup3.hip"), "-o", str(tmp_path / "l3.o")], check=True, timeout=300)
    with pytest.raises(nz.NexusHipError, match="logup column"):
        nz.logup_program_source(frac, AE.RELATION_COLS + 12, 2)
    bad = ap.Program(np.concatenate([frac.instrs[-1:], frac.instrs[:-1]]), frac.n_regs, frac.econsts, 0)     # the last fraction first: batches out of order
    with pytest.raises(nz.NexusHipError, match="batch order"):
        nz.logup_program_source(bad, AE.RELATION_COLS + 12, 3)
    with pytest.raises(nz.NexusHipError, match="constraint instruction"):
        nz.logup_program_source(pb.build(), AE.RELATION_COLS + 12, 3)
