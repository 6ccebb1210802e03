"""CPU tests of the reference-shaped machine's oracle statement (tests/machine_ref.py): the logup interaction trace satisfies the
recorded logup constraints (the oracle prover's own OODS check, ProvingError::ConstraintsNotSatisfied, would fire otherwise), the
oracle's independent verifier session accepts the proof and rejects tampering.  The -m gpu suite then compares nx_prove_machine
with these bytes."""
import numpy as np
import pytest

import machine_ref as M
import oracle_lib as O

P = O.P


def _verify(comps, cfg, words, ad=b""):
    """core::verifier::verify with the transcript prefix of reference machine.rs:299-485 for the machine."""
    import ref_emitter as ap
    hdr = 6
    roots = [words[hdr + 8 * t: hdr + 8 * (t + 1)] for t in range(3)]
    v = O.VerifierSession(cfg)
    for byte in ad:
        v.mix_u64(byte)
    for c in comps:
        v.mix_u64(c[0])
    v.commit(roots[0], [c[0] for c in comps for _ in range(c[1])])
    v.commit(roots[1], [c[0] for c in comps for _ in range(c[2])])
    z, alpha = v.draw_felts(2)
    # the claimed sums travel next to the proof in the reference (Proof.claimed_sum, machine.rs:93-98); here they are recomputed
    # from the prover's side by the caller and passed in through `claimed`
    return v, z, alpha, roots


@pytest.mark.parametrize("comps,kw", [
    ([(6, 3, 9, 8)], dict(pow_bits=4)),
    ([(7, 2, 20, 12), (5, 2, 4, 4), (4, 2, 3, 0)], dict(pow_bits=3, log_constraint_degree=2)),
    ([(5, 2, 5, 4)], dict(pow_bits=2, hash_mode=1)),
    # per-component constraint-degree bounds (the reference's are per component): big +1 components, one small +2 component
    ([(7, 2, 20, 12, 1), (5, 2, 4, 4, 2), (4, 2, 3, 0, 1)], dict(pow_bits=3, log_constraint_degree=2)),
    ([(6, 2, 9, 8, 2), (6, 2, 4, 4, 1), (3, 2, 3, 0)], dict(pow_bits=3, log_constraint_degree=2)),     # the v1 shape: main +2, extensions +1, one defaulted
    # the reference's other logup forms: finalize_logup_in_pairs (two fractions per column, degree 3 under the bound +1),
    # an odd number of fractions, and table components whose tuples read preprocessed columns with -multiplicity numerators
    ([(6, 3, 9, 12, 1, M.PAIRS)], dict(pow_bits=3)),
    ([(6, 3, 9, 8, 0, M.PAIRS | M.ODD), (5, 4, 3, 4, 1, M.TABLE), (4, 5, 4, 8, 1, M.TABLE | M.PAIRS)], dict(pow_bits=3)),
    ([(6, 2, 9, 8, 2, 0), (5, 2, 6, 4, 1, M.PAIRS | M.ODD), (5, 3, 4, 12, 1, M.PAIRS | M.TABLE | M.ODD)], dict(pow_bits=3, log_constraint_degree=2)),
])
def test_oracle_machine_proves_and_verifies(oracle, comps, kw):
    import ref_emitter as ap
    cfg = O.default_cfg(**kw)
    ad = b"\x05"
    words = M.prove_machine(comps, cfg, seed=11, ad=ad, threads=4)
    # deterministic
    assert np.array_equal(words, M.prove_machine(comps, cfg, seed=11, ad=ad, threads=2))
    # verify: rebuild the statement on the verifier side (claimed sums recomputed from the trace, as the reference ships them with the proof)
    v, z, alpha, roots = _verify(comps, cfg, words, ad)
    main, pre = O.synth_tree_columns(comps, 1, 11), O.synth_tree_columns(comps, 0, 11)
    claimed, shifts, off, poff = [], [], 0, 0
    for c in comps:
        _, cs = M.interaction_trace(c, main[off:off + c[2]], z, alpha, pre[poff:poff + c[1]])
        off += c[2]; poff += c[1]
        claimed.append(cs)
        n_inv = pow((1 << c[0]) % P, P - 2, P)
        shifts.append(np.array([(int(x) * n_inv) % P for x in cs], np.uint32))
    v.mix_felts(np.array(claimed, np.uint32))
    v.commit(roots[2], [c[0] for c in comps for _ in range(c[3])])
    locs, a, b, d = [], 0, 0, 0
    for c in comps:
        locs.append((a, b, d)); a += c[1]; b += c[2]; d += c[3]
    components = [M.machine_component(ap, c, l, z, alpha, sh, int(cfg[6])) for c, l, sh in zip(comps, locs, shifts)]
    assert v.verify(components, words) is None
    bad = words.copy(); bad[len(bad) // 2] ^= 1
    v2, z2, a2, _ = _verify(comps, cfg, bad, ad)
    v2.mix_felts(np.array(claimed, np.uint32)); v2.commit(roots[2], [c[0] for c in comps for _ in range(c[3])])
    assert v2.verify(components, bad) is not None


@pytest.mark.parametrize("mode", [0, M.PAIRS, M.PAIRS | M.ODD, M.TABLE | M.PAIRS, M.TUPLES(M.V1), M.PAIRS | M.TUPLES(M.KECCAK), M.PAIRS | M.ODD | M.TUPLES(M.V2)])
def test_logup_constraints_catch_a_wrong_interaction_trace(oracle, mode):
    """A trace built from other tuple columns than the AIR constrains breaks the recorded logup constraints — in every logup form: the
    oracle prover's OODS check refuses."""
    import ref_emitter as ap
    comps = [(5, 4, 6, 8, 1, mode)]
    cfg = O.default_cfg(pow_bits=2)
    real = M.frac_shape

    def shifted(comp, f):
        tree, ent, num, m = real(comp, f)
        k = next(i for i, e in enumerate(ent) if e[0] != "const")                               # the first entry that reads a column: the next column instead
        return tree, ent[:k] + [(ent[k][0], (ent[k][1] + 1) % comp[1 + tree]) + tuple(ent[k][2:])] + ent[k + 1:], num, m
    try:
        M.frac_shape = shifted                                                                   # trace built for other tuple columns ...
        main, pre = O.synth_tree_columns(comps, 1, 3), O.synth_tree_columns(comps, 0, 3)
        s = O.ProverSession(cfg, 5, 2)
        s.mix_u64(5)
        s.commit(pre); s.commit(main)
        z, alpha = s.draw_felts(2)
        cols, cs = M.interaction_trace(comps[0], main, z, alpha, pre)
        s.mix_felts(np.array([cs], np.uint32)); s.commit(cols)
        M.frac_shape = real                                                                      # ... than the AIR constrains
        n_inv = pow(32, P - 2, P)
        comp = M.machine_component(ap, comps[0], (0, 0, 0), z, alpha, np.array([(int(x) * n_inv) % P for x in cs], np.uint32))
        with pytest.raises(RuntimeError):
            s.prove([comp])
    finally:
        M.frac_shape = real


def test_per_component_degree_bound_sets_the_composition_size(oracle):
    """The composition polynomial's log size is the maximum over components of log_size + bound (reference
    prover2/machine/src/prove.rs:44-48; v1: main +2, extensions +1 — prover/src/components/mod.rs:12, extensions/multiplicity.rs:108-110):
    with big +1 components and one small +2 component the composition tree is HALF the size a global bound of 2 gives — other roots,
    fewer decommitment hashes, a shorter proof — and a bound above the configuration's is refused."""
    big = [(7, 2, 20, 12), (5, 2, 4, 4), (4, 2, 3, 0)]
    cfg2 = O.default_cfg(pow_bits=3, log_constraint_degree=2)
    per = [big[0] + (1,), big[1] + (2,), big[2] + (1,)]
    w_global = M.prove_machine(big, cfg2, seed=4, ad=b"d", threads=4)
    w_per = M.prove_machine(per, cfg2, seed=4, ad=b"d", threads=4)
    w_default = M.prove_machine([c + (0,) for c in big], cfg2, seed=4, ad=b"d", threads=4)
    assert np.array_equal(w_global, w_default)                       # 0 = the configuration's bound
    assert len(w_per) < len(w_global)                                # composition tree 2^(8+1) instead of 2^(9+1) leaves
    assert np.array_equal(w_per[6:30], w_global[6:30])               # the three trace roots do not depend on the bound ...
    assert not np.array_equal(w_per[30:38], w_global[30:38])         # ... the composition root does
    # the synthetic machine of nx_prove_synth takes the same field
    sy_g = O.prove_synth([c for c in big], cfg2, seed=9)
    sy_p = O.prove_synth(per, cfg2, seed=9)
    assert len(sy_p) < len(sy_g) and O.verify_synth(per, cfg2, sy_p) is None and O.verify_synth(big, cfg2, sy_p) is not None


@pytest.mark.parametrize("comps,kw", [
    ([(6, 3, 9, 8)], dict(pow_bits=2)),
    ([(6, 3, 9, 12, 1, M.PAIRS), (5, 2, 6, 8, 1, M.PAIRS | M.ODD)], dict(pow_bits=2)),
    ([(6, 4, 9, 20, 0, M.PAIRS | M.ODD), (5, 4, 3, 4, 1, M.TABLE), (4, 5, 4, 8, 1, M.TABLE | M.PAIRS)], dict(pow_bits=2)),
    ([(6, 2, 19, 36, 2, M.PAIRS), (5, 3, 4, 12, 1, M.PAIRS | M.TABLE | M.ODD), (5, 3, 4, 8, 2, M.TABLE)], dict(pow_bits=2, log_constraint_degree=2)),
    # the reference's tuple widths and entry kinds (VERDICT r5 missing #3): 1 / 3 / 4 / 9 with a constant, a sum of two columns and a flag-column
    # numerator (v1), 3 / 4 and two 200-wide state fractions with the numerators m - 1 and 1 - m (keccak), 9 ... 21 (prover2)
    ([(6, 3, 12, 36, 0, M.TUPLES(M.V1))], dict(pow_bits=2)),
    ([(6, 3, 12, 16, 1, M.PAIRS | M.TUPLES(M.KECCAK)), (5, 4, 6, 8, 1, M.TABLE | M.TUPLES(M.KECCAK))], dict(pow_bits=2)),
    ([(6, 3, 30, 20, 1, M.PAIRS | M.ODD | M.TUPLES(M.V2)), (5, 2, 8, 12, 2, M.TUPLES(M.V1))], dict(pow_bits=2, log_constraint_degree=2)),
])
def test_product_emission_equals_the_checkers_proof(oracle, comps, kw):
    """The PRODUCT's recorded program for a machine component (csrc/machine.hip machine_component, handed out by the host-only export
    nx_machine_air_program) run through the ORACLE's prover session gives the proof of the checker's own, independent emission
    (tests/ref_emitter.py) — in every logup form.  No GPU: the -m gpu suite then compares the device prover with these bytes."""
    import ctypes as C
    import nexus_zkvm_amd as nz
    import ref_emitter as RE
    lib = nz.load_library()

    def product_component(ap, comp, loc, z, alpha, shift, cfg_lcd):
        chk = M.machine_component(ap, comp, loc, z, alpha, shift, cfg_lcd)       # the column list and the declared masks are the statement's
        spec = nz.ComponentSpec(*[int(x) for x in comp])
        prog, n, regs, nc = C.c_void_p(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        assert lib.nx_machine_air_program(C.byref(spec), cfg_lcd, C.byref(prog), C.byref(n), C.byref(regs), C.byref(nc)) == 0
        ins = np.ctypeslib.as_array(C.cast(prog, C.POINTER(C.c_uint32)), shape=(n.value, 4)).copy()
        lib.nx_free_host(prog)
        assert nc.value == chk.program.n_constraints
        econsts = [list(map(int, z)), list(map(int, alpha)), list(map(int, shift)), [0, 0, 0, 0]]
        if M.tuple_sched(comp):                                   # a wide-tuple component's E constants go on with alpha^0, alpha^1, ... (include/nexus_hip.h NX_LOGUP_TUPLES)
            pw = np.array([1, 0, 0, 0], np.uint32)
            for _ in range(max(len(M.frac_shape(comp, f)[1]) for f in range(M.n_fracs(comp)))):
                econsts.append(list(map(int, pw))); pw = O.qm31_mul(pw, alpha)
        p = RE.Program([tuple(int(x) for x in row) for row in ins], econsts, regs.value, nc.value, {})
        return RE.Component(chk.log_size, p, chk.cols, chk.masks, chk.log_constraint_degree_bound)

    cfg = O.default_cfg(**kw)
    ref = M.prove_machine(comps, cfg, seed=21, ad=b"e", threads=4)
    got = M.prove_machine(comps, cfg, seed=21, ad=b"e", threads=4, component_fn=product_component)
    assert np.array_equal(ref, got)
