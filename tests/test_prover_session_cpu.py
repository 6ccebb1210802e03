"""The oracle's generic prove / verify sessions (oracle/air_generic.h): stwo::prover::prove over RECORDED AIRs.

Pins: (1) the synthetic machine expressed as a recorded component, driven through the session with the reference's transcript
prefix (machine.rs:198-263), yields byte-for-byte the proof of the built-in prove_synth — two independent statements of the
composition polynomial, mask points and OODS check agree; (2) prove -> verify round trips on a multi-component AIR with
components of different sizes, a logup-style secure column read at offsets [-1, 0] and secure constants; (3) the verifier
rejects tampered proofs, wrong lookup elements and a wrong transcript; the prover refuses a trace that violates a constraint."""
import numpy as np
import pytest

import oracle_lib as O
import air_examples as X

P = O.P


def _ap():
    import nexus_zkvm_amd.air_program as ap
    return ap


def drive_synthetic(session_cls_args, comps, seed, ad, commit, channel):
    """machine.rs:198-263 for the synthetic machine: returns nothing, leaves three trees committed."""
    for b in ad:
        channel.mix_u64(b)
    for c in comps:
        channel.mix_u64(c[0])
    commit(O.synth_tree_columns(comps, 0, seed))
    commit(O.synth_tree_columns(comps, 1, seed))
    z = channel.draw_felt()
    inter_seed = O.lib().orc_inter_seed_from(O.ptr(z))
    channel.mix_felts(np.zeros((len(comps), 4), np.uint32))
    commit(O.synth_tree_columns(comps, 2, seed, inter_seed))


def synthetic_components(comps):
    ap = _ap()
    out, a, b, c = [], 0, 0, 0
    for (log, n_pre, n_main, n_inter) in comps:
        out.append(X.synthetic_component(ap, log, n_pre, n_main, n_inter, a, b, c))
        a, b, c = a + n_pre, b + n_main, c + n_inter
    return out


@pytest.mark.parametrize("comps,lcd", [([(6, 3, 20, 19)], 1), ([(5, 2, 18, 4), (7, 3, 5, 17)], 2), ([(6, 2, 3, 0)], 1)])
def test_recorded_synthetic_machine_gives_the_built_in_proof(comps, lcd):
    cfg = O.default_cfg(pow_bits=3, log_constraint_degree=lcd, log_blowup=max(1, lcd))
    ad = b"\x07\x2a"
    ref = O.prove_synth(comps, cfg, seed=9, ad=ad)
    s = O.ProverSession(cfg, max(c[0] for c in comps))
    drive_synthetic(None, comps, 9, ad, s.commit, s)
    air = synthetic_components(comps)
    words = s.prove(air)
    assert np.array_equal(words, ref)
    # and the generic verifier accepts it, replaying the same prefix
    v = O.VerifierSession(cfg)
    roots = _roots_from_proof(words)
    for b in ad:
        v.mix_u64(b)
    for c in comps:
        v.mix_u64(c[0])
    v.commit(roots[0], [c[0] for c in comps for _ in range(c[1])])
    v.commit(roots[1], [c[0] for c in comps for _ in range(c[2])])
    v.draw_felt()
    v.mix_felts(np.zeros((len(comps), 4), np.uint32))
    v.commit(roots[2], [c[0] for c in comps for _ in range(c[3])])
    assert v.verify(air, words) is None
    assert O.verify_synth(comps, cfg, words, ad=ad) is None


def _roots_from_proof(words):
    """NXP1 layout (oracle/pcs.h::proof_serialize): magic, 7 config words?, ... — located by searching is brittle, so the
    tests take the roots from the prover session instead when they can; here they are parsed: header then 4 commitments."""
    w = np.asarray(words, np.uint32)
    n_hdr = O.proof_header_words()
    assert w[n_hdr] == 4
    return [w[n_hdr + 1 + 8 * t: n_hdr + 9 + 8 * t].copy() for t in range(4)]


def build_mixed_air(logs=(5, 7), seed=3, lcd=1, bounds=None, high_degree=False):
    """Two logup components of different sizes + the committed columns.  Returns (drive(session) -> components, tree_logs).
    bounds: per-component log constraint-degree bounds (0 / None = the configuration's).  high_degree: the components whose bound is
    2 also carry degree-3 and degree-4 constraints (base- and secure-field), interleaved with the degree-2 ones."""
    ap = _ap()

    def drive(sess, commit, tamper=None):
        sess.mix_u64(len(logs))
        mains = [X.logup_main_trace(l, seed + i) for i, l in enumerate(logs)]
        if tamper == "main":
            mains[0][1][2][5] = (int(mains[0][1][2][5]) + 1) % P
        commit([np.zeros(1 << logs[0], np.uint32)])                                # a (dummy) preprocessed column
        commit([c for _, fin in mains for c in fin])
        z, alpha = sess.draw_felt(), sess.draw_felt()
        inter, comps = [], []
        for i, l in enumerate(logs):
            cols4, shift = X.logup_interaction_trace(l, mains[i][0], z, alpha)
            inter += cols4
            hd = high_degree and ((bounds[i] if bounds and bounds[i] else lcd) >= 2)
            comps.append(X.logup_component(ap, l, z, alpha, shift, main0=3 * i, inter0=4 * i, high_degree=hd))
            comps[-1].log_constraint_degree_bound = bounds[i] if bounds else 0
            sess.mix_felts(shift)
        commit(inter)
        # the preprocessed column is claimed (sampled at 0) by a constraint-free reader: attach it to component 0
        c0 = comps[0]
        comps[0] = ap.Component(c0.log_size, c0.program, c0.cols + [(0, 0)], c0.masks + [[0]], c0.log_constraint_degree_bound)
        return comps
    tree_logs = [[logs[0]], [l for l in logs for _ in range(3)], [l for l in logs for _ in range(4)]]
    return drive, tree_logs


@pytest.mark.parametrize("logs,lcd,bounds,hd", [((5, 7), 1, None, False), ((6,), 2, None, False), ((7, 5, 6), 1, None, False),
                                                ((7, 5, 6), 2, (1, 2, 1), False),      # per-component bounds: composition 2^8, not 2^9
                                                ((6, 6), 2, (2, 1), False),
                                                ((6, 5), 2, (2, 1), True),              # degree-3 / degree-4 constraints under the +2 bound
                                                ((5, 6), 2, None, True)])
def test_logup_air_round_trip_and_rejections(logs, lcd, bounds, hd):
    cfg = O.default_cfg(pow_bits=2, log_constraint_degree=lcd, log_blowup=lcd)
    drive, tree_logs = build_mixed_air(logs, lcd=lcd, bounds=bounds, high_degree=hd)
    s = O.ProverSession(cfg, max(logs))
    roots = []
    comps = drive(s, lambda cols: roots.append(s.commit(cols)))
    words = s.prove(comps)

    def verifier(components, w, extra_mix=None):
        v = O.VerifierSession(cfg)
        v.mix_u64(len(logs) if extra_mix is None else extra_mix)
        v.commit(roots[0], tree_logs[0]); v.commit(roots[1], tree_logs[1])
        z, alpha = v.draw_felt(), v.draw_felt()
        for c in components:
            v.mix_felts(np.asarray(c.program.econsts, np.uint32)[2])
        v.commit(roots[2], tree_logs[2])
        return v.verify(components, w), (z, alpha)

    err, (z, alpha) = verifier(comps, words)
    assert err is None
    assert np.array_equal(np.asarray(comps[0].program.econsts, np.uint32)[0], z)
    bad = words.copy(); bad[len(bad) // 2] ^= 1
    assert verifier(comps, bad)[0] is not None
    assert verifier(comps, words, extra_mix=99)[0] is not None                     # different transcript
    # a verifier that assumes other lookup elements evaluates other constraints: OODS mismatch
    ap = _ap()
    other = [X.logup_component(ap, l, (1, 2, 3, 4), alpha, np.asarray(c.program.econsts, np.uint32)[2], 3 * i, 4 * i, high_degree=c.program.n_constraints > 2)
             for i, (l, c) in enumerate(zip(logs, comps))]
    for o, c in zip(other, comps):
        o.log_constraint_degree_bound = c.log_constraint_degree_bound
    other[0] = ap.Component(other[0].log_size, other[0].program, other[0].cols + [(0, 0)], other[0].masks + [[0]], other[0].log_constraint_degree_bound)
    assert "Oods" in verifier(other, words)[0]


def test_prover_refuses_an_invalid_trace_and_malformed_airs():
    cfg = O.default_cfg(pow_bits=2)
    drive, tree_logs = build_mixed_air((5,))
    s = O.ProverSession(cfg, 5)
    comps = drive(s, s.commit, tamper="main")
    with pytest.raises(RuntimeError, match="ConstraintsNotSatisfied"):
        s.prove(comps)
    ap = _ap()
    s = O.ProverSession(cfg, 5)
    comps = drive(s, s.commit)
    c0 = comps[0]
    with pytest.raises(RuntimeError, match="claimed by no component"):
        s.prove([ap.Component(c0.log_size, c0.program, c0.cols[:-1], c0.masks[:-1])])
    s = O.ProverSession(cfg, 5)
    comps = drive(s, s.commit)
    with pytest.raises(RuntimeError, match="missing from the column's mask"):
        s.prove([ap.Component(c0.log_size, c0.program, comps[0].cols, [[0]] * len(comps[0].cols))])
    s = O.ProverSession(cfg, 5)
    s.commit([np.zeros(32, np.uint32)])
    with pytest.raises(RuntimeError, match="outside the committed trees"):          # one tree committed, the components name three
        s.prove(comps)


@pytest.mark.parametrize("n_trees", [2, 3, 4])
def test_sessions_take_the_tree_count_they_are_given(n_trees):
    """VERDICT r3 hygiene: the prover hard-coded "exactly three trace trees".  The reference commits three (machine.rs:208-263) but
    Stwo's prove takes any TreeVec: the session now proves whatever was committed — here 2 trees (no interaction tree), 3, and 4 (some
    main columns in a fourth tree) — and the verifier session accepts exactly that statement (and not a proof of another tree count)."""
    ap = _ap()
    cfg = O.default_cfg(pow_bits=3)
    trees, comp = X.tree_count_statement(ap, n_trees)
    s = O.ProverSession(cfg, comp.log_size)
    s.mix_u64(n_trees)
    roots = [s.commit(t) for t in trees]
    words = s.prove([comp])
    assert int(words[O.proof_header_words()]) == n_trees + 1
    v = O.VerifierSession(cfg)
    v.mix_u64(n_trees)
    for r, t in zip(roots, trees):
        v.commit(r, [comp.log_size] * len(t))
    assert v.verify([comp], words) is None
    bad = words.copy(); bad[len(bad) // 2] ^= 1
    v2 = O.VerifierSession(cfg)
    v2.mix_u64(n_trees)
    for r, t in zip(roots, trees):
        v2.commit(r, [comp.log_size] * len(t))
    assert v2.verify([comp], bad) is not None
    if n_trees > 2:                                    # a verifier that saw fewer trees refuses the proof's structure
        v3 = O.VerifierSession(cfg)
        v3.mix_u64(n_trees)
        for r, t in list(zip(roots, trees))[:-1]:
            v3.commit(r, [comp.log_size] * len(t))
        assert v3.verify([comp], words) is not None
