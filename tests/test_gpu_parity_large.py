"""GPU parity at the sizes the bench runs (-m gpu): byte-for-byte against the CPU oracle, not just verifier acceptance.

VERDICT r1 "what's weak" #2: a sparse wrong-row bug above 2^16 rows (an index overflow, a tile that is skipped) passes a 3-query
verifier with high probability.  These tests compare WHOLE outputs at BASELINE sizes:
  * config #2 as specified: all 347 columns x 2^20 rows, uniform values and the byte-limb variant
    (values in [0, 256), reference prover/src/trace/utils.rs:57-62) — every coefficient, every LDE value, the root;
  * K7 / K8 / K9 standalone at 2^20 / 2^21 rows;
  * the full 27 / 347 / 64 machine at 2^20 rows: every proof word;
  * (slow, NX_RUN_SLOW=1) the headline 2^22-row proof, every word.
The oracle runs on all host cores; each test is sized to finish in well under a minute of host work.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch  # noqa: F401  (HIP runtime load order, see test_gpu_parity.py)

import oracle_lib as O

pytestmark = pytest.mark.gpu
P = O.P
THREADS = max(4, os.cpu_count() or 4)


@pytest.fixture(scope="module")
def be():
    import nexus_zkvm_amd as nz
    b = nz.HipBackend(0)
    yield b
    b.close()


@pytest.fixture(scope="module")
def nz():
    import nexus_zkvm_amd
    return nexus_zkvm_amd


def _rand_point_alpha(seed):
    L = O.lib()
    ch = C.c_void_p(L.orc_channel_new())
    L.orc_channel_mix_u64(ch, seed)
    p1, p2, a = np.zeros(8, np.uint32), np.zeros(8, np.uint32), np.zeros(4, np.uint32)
    L.orc_get_random_point(ch, O.ptr(p1))
    L.orc_get_random_point(ch, O.ptr(p2))
    L.orc_channel_draw_secure_felt(ch, O.ptr(a))
    L.orc_channel_free(ch)
    return p1, p2, a


@pytest.mark.parametrize("variant", ["uniform", "byte_limbs"])
def test_config2_all_347_columns_bit_exact(be, oracle, variant):
    """BASELINE config #2 exactly as SURVEY §8(d) states it: C = 347 columns, n = 20, SplitMix-style seeded values (uniform in [0, P),
    or byte limbs in [0, 256) like the reference's limb columns) -> R3 is not involved (inputs are already bit-reversed evaluations)
    -> K3 -> K4 -> K5.  Coefficients, LDE values and the Merkle root of nx_lde_commit against the oracle, all 347 columns."""
    log, n_cols = 20, 347
    rng = np.random.default_rng(0xC0FFEE)
    hi = P if variant == "uniform" else 256
    vals = rng.integers(0, hi, (n_cols, 1 << log), dtype=np.uint32)
    tw = be.precompute_twiddles(log)
    cols = be.columns_from_host(vals)
    lde, root = be.lde_commit(tw, cols, 1)
    ref_cols = [np.ascontiguousarray(v).copy() for v in vals]
    ref_lde, ref_root = oracle.lde_commit(ref_cols, 1, threads=THREADS)
    assert np.array_equal(root, ref_root), variant
    got_c = cols.to_cpu()
    for c in range(n_cols):
        assert np.array_equal(got_c[c], ref_cols[c]), (variant, "coefficients", c)
    del got_c
    got_l = lde.to_cpu()
    assert got_l.max() < P
    for c in range(n_cols):
        assert np.array_equal(got_l[c], ref_lde[c]), (variant, "lde", c)
    lde.free(); cols.free()


def test_eval_at_point_2pow20(be, oracle):
    """K7 at the bench's polynomial size and above the 2^15 the small suite stops at."""
    for log in (20, 22):
        polys = np.random.default_rng(log).integers(0, P, (3, 1 << log), dtype=np.uint32)
        d = be.columns_from_host(polys)
        p1, p2, _ = _rand_point_alpha(log)
        idx, pts = [0, 1, 2, 1], [p1, p1, p1, p2]
        got = be.eval_at_points(d, idx, pts)
        for i, (pi, pt) in enumerate(zip(idx, pts)):
            assert np.array_equal(got[i], oracle.eval_at_point(polys[pi], pt)), (log, i)
        d.free()


def test_accumulate_quotients_2pow21(be, oracle):
    """K8 on the LDE domain of a 2^20-row trace, two sample batches (the mask-[0, 1] shape), every output word."""
    log, n_cols = 21, 40
    cols = np.random.default_rng(77).integers(0, P, (n_cols, 1 << log), dtype=np.uint32)
    p1, p2, alpha = _rand_point_alpha(5)
    rng = np.random.default_rng(6)
    b1 = [(c, rng.integers(0, P, 4, dtype=np.uint32)) for c in range(n_cols)]
    b2 = [(c, rng.integers(0, P, 4, dtype=np.uint32)) for c in (0, 1)]
    d = be.columns_from_host(cols)
    got = be.accumulate_quotients(d, alpha, [(p1, b1), (p2, b2)]).to_cpu()
    outs = [np.zeros(1 << log, np.uint32) for _ in range(4)]
    pts = np.concatenate([p1, p2])
    counts = np.array([len(b1), len(b2)], np.int32)
    cidx = np.array([c for c, _ in b1 + b2], np.int32)
    vals = np.concatenate([v for _, v in b1 + b2])
    oracle.lib().orc_accumulate_quotients(log, O.ptr_array([np.ascontiguousarray(c) for c in cols]), n_cols, O.ptr(alpha), 2, O.ptr(pts),
                                          O.ptr(counts), O.ptr(cidx), O.ptr(vals), THREADS, O.ptr_array(outs))
    assert np.array_equal(got, np.stack(outs))
    d.free()


def test_fri_folds_2pow21(be, oracle):
    """K9 on the first FRI layers of a 2^20-row proof: circle -> line at 2^21, line folds at 2^20 (first and a doubled domain)."""
    L = oracle.lib()
    log = 21
    tw = be.precompute_twiddles(log)
    src = np.random.default_rng(91).integers(0, P, (4, 1 << log), dtype=np.uint32)
    dst0 = np.random.default_rng(92).integers(0, P, (4, 1 << (log - 1)), dtype=np.uint32)
    _, _, alpha = _rand_point_alpha(17)
    d_src, d_dst = be.columns_from_host(src), be.columns_from_host(dst0)
    be.fold_circle_into_line(tw, d_dst, d_src, alpha)
    ref = [np.ascontiguousarray(c).copy() for c in dst0]
    L.orc_fold_circle_into_line(O.ptr_array(ref), O.ptr_array([np.ascontiguousarray(c) for c in src]), log, O.ptr(alpha))
    assert np.array_equal(d_dst.to_cpu(), np.stack(ref))
    line = np.ascontiguousarray(src[:, :1 << (log - 1)])
    d_line = be.columns_from_host(line)
    for dbl in (0, 1):
        out = be.fold_line(tw, d_line, alpha, dbl).to_cpu()
        ref2 = [np.zeros(1 << (log - 2), np.uint32) for _ in range(4)]
        L.orc_fold_line_dom(O.ptr_array([np.ascontiguousarray(c) for c in line]), log - 1, dbl, O.ptr(alpha), O.ptr_array(ref2))
        assert np.array_equal(out, np.stack(ref2)), dbl


def _assert_same_proof(ref, words):
    assert len(ref) == len(words)
    if not np.array_equal(ref, words):
        bad = int(np.nonzero(ref != words)[0][0])
        pytest.fail(f"first differing proof word {bad} of {len(ref)} (roots are words 6..37)")


@pytest.mark.parametrize("kw", [dict(), dict(hash_mode=1)])
def test_whole_proof_byte_equal_at_2pow20_full_machine(be, nz, oracle, kw):
    """The machine the bench proves (27 preprocessed + 347 main + 64 interaction columns) at 2^20 rows — the size of bench.py's
    cpu_baseline sample: every proof word equals the oracle's, in both Merkle hash rules."""
    comps = [(20, 27, 347, 64)]
    words = be.prove(comps, nz.default_config(**kw), seed=7)
    ref = oracle.prove_synth(comps, O.default_cfg(**kw), seed=7, threads=THREADS)
    _assert_same_proof(ref, words)


@pytest.mark.skipif(os.environ.get("NX_RUN_SLOW", "0") != "1", reason="slow: the oracle proves 2^22 rows x 438 columns on the host (minutes, ~40 GB of RAM); NX_RUN_SLOW=1")
def test_whole_proof_byte_equal_at_2pow22_headline(be, nz, oracle):
    """BASELINE config #3 itself: the headline 2^22-row proof, every word, against the oracle."""
    comps = [(22, 27, 347, 64)]
    words = be.prove(comps, nz.default_config(), seed=2001)
    ref = oracle.prove_synth(comps, O.default_cfg(), seed=2001, threads=THREADS)
    _assert_same_proof(ref, words)


def test_config4_trace_2pow24_lde_matches_oracle(be, oracle):
    """BASELINE config #4's trace size on ONE GPU: nx_lde_batch of 2^24-row columns onto 2^25 points — the three-pass transform plan
    (13 + 6 + 5 layers, the fused middle launch at its smallest layer count): coefficients and every LDE value of two columns
    against the oracle (values at both ends of the field in the first rows)."""
    log = 24
    vals = np.random.default_rng(2400).integers(0, P, (2, 1 << log), dtype=np.uint32)
    vals[0, :4] = [P - 1, 0, P - 1, 1]
    tw = be.precompute_twiddles(log)
    otw = oracle.Twiddles(log + 1)
    cols = be.columns_from_host(vals)
    lde = be.lde(tw, cols, 1)
    coeffs = np.stack([otw.interpolate(v) for v in vals])
    assert np.array_equal(cols.to_cpu(), coeffs)
    got = lde.to_cpu()
    assert got.max() < P
    for c in range(2):
        assert np.array_equal(got[c], otw.evaluate(coeffs[c], log + 1)), c
    lde.free(); cols.free()


def test_config4_trace_2pow24_prove_is_accepted_and_tamper_rejected(be, nz, oracle):
    """A 2^24-row statement (27 + 347 + 64 columns: BASELINE config #4's trace, proved on one GPU — 288 GB of HBM hold its 110 GB of
    coefficients and LDE): the oracle's verifier (whose cost does not grow with the trace) accepts the proof, rejects a flipped word
    and a different transcript.  Size-independent property check; byte parity with the oracle prover stops at 2^22 (NX_RUN_SLOW)."""
    comps = [(24, 27, 347, 64)]
    cfg, ocfg = nz.default_config(pow_bits=8), O.default_cfg(pow_bits=8)
    w = be.prove(comps, cfg, seed=24, ad=b"cfg4")
    assert oracle.verify_synth(comps, ocfg, w, ad=b"cfg4") is None
    for pos in (len(w) // 3, len(w) - 7):
        bad = w.copy(); bad[pos] ^= 1
        assert oracle.verify_synth(comps, ocfg, bad, ad=b"cfg4") is not None
    assert oracle.verify_synth(comps, ocfg, w, ad=b"other") is not None
