"""CPU tests pinning the oracle (oracle/) against independent references.

The reference holds no golden vectors for this path ("parity unpinned"), so the
oracle is pinned against (a) RFC 7693 / hashlib for Blake2s, (b) Python big-integer models of the
field tower restated in the reference's spec (specification/zkvm-spec-3.0.pdf §3.1), (c) the
reference's only layout test `test_order` (prover/src/trace/utils.rs:117-128), and (d)
mathematical identities of the Circle FFT / DEEP quotients / FRI folds.
"""
import ctypes as C
import hashlib
import json
import os
import random

import numpy as np
import pytest

import oracle_lib as O

P = O.P
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---------- independent Python model of the field tower / circle group ----------
def cm_mul(x, y):
    return ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def qm_mul(x, y):
    a, b, c, d = (x[0], x[1]), (x[2], x[3]), (y[0], y[1]), (y[2], y[3])
    ac, bd = cm_mul(a, c), cm_mul(b, d)
    rbd = cm_mul(bd, (2, 1))
    ad, bc = cm_mul(a, d), cm_mul(b, c)
    return ((ac[0] + rbd[0]) % P, (ac[1] + rbd[1]) % P, (ad[0] + bc[0]) % P, (ad[1] + bc[1]) % P)


def pt_add(p, q):
    return ((p[0] * q[0] - p[1] * q[1]) % P, (p[0] * q[1] + p[1] * q[0]) % P)


def pt_mul(p, k):
    r = (1, 0)
    while k:
        if k & 1:
            r = pt_add(r, p)
        p = pt_add(p, p)
        k >>= 1
    return r


GEN = (2, 1268011823)


def test_m31_arithmetic(oracle):
    L = oracle.lib()
    rnd = random.Random(1)
    edge = [0, 1, 2, P - 1, P - 2, 1 << 30, (1 << 30) + 1, 65535, 65536]
    vals = edge + [rnd.randrange(P) for _ in range(200)]
    for a in vals:
        for b in vals[:40]:
            assert L.orc_m31_mul(a, b) == a * b % P
            assert L.orc_m31_add(a, b) == (a + b) % P
            assert L.orc_m31_sub(a, b) == (a - b) % P
        if a:
            assert L.orc_m31_mul(L.orc_m31_inv(a), a) == 1
    for x in [0, P, P * P - 1, (P - 1) * (P - 1), 1 << 31, P * P - P] + [rnd.randrange(P * P) for _ in range(200)]:
        assert L.orc_m31_reduce(x) == x % P


def test_qm31_arithmetic(oracle):
    L = oracle.lib()
    rnd = random.Random(2)
    for _ in range(200):
        a = [rnd.randrange(P) for _ in range(4)]
        b = [rnd.randrange(P) for _ in range(4)]
        out = np.zeros(4, np.uint32)
        L.orc_qm31_mul(O.ptr(O.u32(a)), O.ptr(O.u32(b)), O.ptr(out))
        assert tuple(int(x) for x in out) == qm_mul(a, b)
        inv = np.zeros(4, np.uint32)
        L.orc_qm31_inv(O.ptr(O.u32(a)), O.ptr(inv))
        assert qm_mul(a, [int(x) for x in inv]) == (1, 0, 0, 0)
    # u^2 = 2 + i
    assert qm_mul((0, 0, 1, 0), (0, 0, 1, 0)) == (2, 1, 0, 0)


def test_circle_group(oracle):
    L = oracle.lib()
    assert (GEN[0] ** 2 + GEN[1] ** 2) % P == 1
    assert pt_mul(GEN, 1 << 30) == (P - 1, 0)
    assert pt_mul(GEN, 1 << 31) == (1, 0)
    xy = np.zeros(2, np.uint32)
    for idx in [0, 1, 5, 1 << 20, (1 << 31) - 1, 123456789]:
        L.orc_circle_point(idx, O.ptr(xy))
        assert (int(xy[0]), int(xy[1])) == pt_mul(GEN, idx)
    # CanonicCoset(n).circle_domain(): first half = half_odds(n-1), second half = conjugates
    n = 5
    for i in range(1 << n):
        L.orc_circle_domain_at(n, i, O.ptr(xy))
        half = 1 << (n - 1)
        j = i if i < half else i - half
        p = pt_mul(GEN, (1 << (31 - (n - 1) - 2)) + j * (1 << (31 - (n - 1))))
        if i >= half:
            p = (p[0], (-p[1]) % P)
        assert (int(xy[0]), int(xy[1])) == p
    # coset order <-> circle-domain order (reference utils_external.rs:24-39)
    dom = {}
    for i in range(1 << n):
        L.orc_circle_domain_at(n, i, O.ptr(xy))
        dom[i] = (int(xy[0]), int(xy[1]))
    for c in range(1 << n):
        L.orc_canonic_coset_at(n, c, O.ptr(xy))
        assert dom[L.orc_coset_index_to_circle_domain_index(c, n)] == (int(xy[0]), int(xy[1]))


def test_reference_test_order(oracle):
    """Restates the reference's only layout pin: prover/src/trace/utils.rs:117-128 (test_order)."""
    L = oracle.lib()
    for log_size in (3, 4, 7):
        vals = np.arange(1 << log_size, dtype=np.uint32)
        col = oracle.finalize_column(vals)
        for i in range(1 << log_size):
            idx = L.orc_bit_reverse_index(L.orc_coset_index_to_circle_domain_index(i, log_size), log_size)
            assert col[i] == vals[idx]


def test_blake2s_rfc7693_and_hashlib(oracle):
    L = oracle.lib()
    out = (C.c_uint8 * 32)()
    # RFC 7693 Appendix B: BLAKE2s-256("abc")
    L.orc_blake2s(b"abc", 3, out)
    assert bytes(out).hex() == "508c5e8c327c14e2e1a72ba34eeb452f37458b209ed63a294d999b4c86675982"
    rnd = random.Random(3)
    for n in [0, 1, 55, 63, 64, 65, 127, 128, 129, 1000, 4096]:
        data = bytes(rnd.randrange(256) for _ in range(n))
        L.orc_blake2s(data, n, out)
        assert bytes(out) == hashlib.blake2s(data).digest()


def test_hash_node_modes(oracle):
    L = oracle.lib()
    rnd = random.Random(4)
    for nvals in [0, 1, 15, 16, 17, 32, 347]:
        for has_children in (False, True):
            vals = O.u32([rnd.randrange(P) for _ in range(nvals)])
            ch = O.u32([rnd.randrange(1 << 32) for _ in range(16)])
            out = np.zeros(8, np.uint32)
            L.orc_hash_node(O.ptr(ch) if has_children else None, O.ptr(vals), nvals, O.HASH_STD, O.ptr(out))
            msg = (ch.tobytes() if has_children else b"") + vals.tobytes()
            assert out.tobytes() == hashlib.blake2s(msg).digest()
            # legacy rule: zero state, raw compressions, t = f = 0
            L.orc_hash_node(O.ptr(ch) if has_children else None, O.ptr(vals), nvals, O.HASH_RAW0, O.ptr(out))
            st = np.zeros(8, np.uint32)
            if has_children:
                L.orc_blake2s_compress(O.ptr(st), O.ptr(ch), 0, 0, 0, 0)
            padded = np.concatenate([vals, np.zeros((-nvals) % 16, np.uint32)])
            for i in range(0, len(padded), 16):
                blk = np.ascontiguousarray(padded[i:i + 16])
                L.orc_blake2s_compress(O.ptr(st), O.ptr(blk), 0, 0, 0, 0)
            assert (out == st).all()


def test_merkle_mixed_degree_matches_python(oracle):
    rnd = np.random.default_rng(5)
    logs = [5, 3, 5, 4, 3, 5]
    cols = [rnd.integers(0, P, 1 << l, dtype=np.uint32) for l in logs]
    root = oracle.merkle_commit(cols)
    # python model: stable sort by size desc, layer k node i = H(children ‖ values of columns of size 2^k)
    order = sorted(range(len(cols)), key=lambda i: -logs[i])
    prev = None
    for log in range(max(logs), -1, -1):
        lc = [cols[i] for i in order if logs[i] == log]
        cur = []
        for i in range(1 << log):
            msg = b""
            if prev is not None:
                msg += prev[2 * i] + prev[2 * i + 1]
            msg += b"".join(int(c[i]).to_bytes(4, "little") for c in lc)
            cur.append(hashlib.blake2s(msg).digest())
        prev = cur
    assert root.tobytes() == prev[0]


def test_channel_matches_python_model(oracle):
    L = oracle.lib()
    ch = C.c_void_p(L.orc_channel_new())
    digest = bytes(32)
    d = np.zeros(8, np.uint32)
    # mix_u64
    L.orc_channel_mix_u64(ch, 0x1122334455667788)
    digest = hashlib.blake2s(digest + (0x1122334455667788).to_bytes(8, "little")).digest()
    L.orc_channel_digest(ch, O.ptr(d))
    assert d.tobytes() == digest
    # mix_root
    root = O.u32(list(range(8)))
    L.orc_channel_mix_root(ch, O.ptr(root))
    digest = hashlib.blake2s(digest + root.tobytes()).digest()
    L.orc_channel_digest(ch, O.ptr(d))
    assert d.tobytes() == digest
    # draw: H(digest ‖ counter(LE, padded to 32) ‖ 0x00), retry unless all words < 2P, reduce mod P
    out = np.zeros(4, np.uint32)
    n_sent = 0
    for _ in range(3):
        L.orc_channel_draw_secure_felt(ch, O.ptr(out))
        while True:
            h = hashlib.blake2s(digest + n_sent.to_bytes(32, "little") + b"\x00").digest()
            n_sent += 1
            ws = [int.from_bytes(h[4 * i:4 * i + 4], "little") for i in range(8)]
            if all(w < 2 * P for w in ws):
                break
        assert [int(x) for x in out] == [w % P for w in ws[:4]]
    # mix_felts resets the draw counter
    f = O.u32([1, 2, 3, 4, 5, 6, 7, 8])
    L.orc_channel_mix_felts(ch, O.ptr(f), 2)
    digest = hashlib.blake2s(digest + f.tobytes()).digest()
    L.orc_channel_digest(ch, O.ptr(d))
    assert d.tobytes() == digest
    # grind: smallest nonce with >= pow_bits trailing zeros of H(digest ‖ nonce) read as LE u128
    nonce = L.orc_channel_grind(ch, 8)
    for cand in range(nonce + 1):
        h = hashlib.blake2s(digest + cand.to_bytes(8, "little")).digest()
        tz = (int.from_bytes(h[:16], "little") | (1 << 128))
        tz = (tz & -tz).bit_length() - 1
        assert (tz >= 8) == (cand == nonce)
    L.orc_channel_free(ch)


def _direct_eval(oracle, coeffs, n, idx_bitrev):
    L = oracle.lib()
    xy = np.zeros(2, np.uint32)
    L.orc_circle_domain_at(n, L.orc_bit_reverse_index(idx_bitrev, n), O.ptr(xy))
    return L.orc_eval_basis_at_m31_point(O.ptr(coeffs), int(np.log2(len(coeffs))), int(xy[0]), int(xy[1]))


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8])
def test_cfft_matches_direct_basis_evaluation(oracle, n):
    """evaluate(c) == Σ_j c_j·y^{j0}x^{j1}π(x)^{j2}… at CircleDomain.at(bitrev(i));
    a twiddle tree of a LARGER coset serves smaller domains through its tail slices."""
    rnd = np.random.default_rng(n)
    tw = oracle.Twiddles(9)
    coeffs = rnd.integers(0, P, 1 << n, dtype=np.uint32)
    ev = tw.evaluate(coeffs, n)
    for i in range(1 << n):
        assert ev[i] == _direct_eval(oracle, coeffs, n, i)
    assert (tw.interpolate(ev) == coeffs).all()
    # LDE: zero-extended coefficients on the blown-up domain
    ev2 = tw.evaluate(coeffs, n + 1)
    ext = np.concatenate([coeffs, np.zeros(1 << n, np.uint32)])
    for i in range(0, 1 << (n + 1), max(1, (1 << (n + 1)) // 16)):
        assert ev2[i] == _direct_eval(oracle, ext, n + 1, i)


def test_eval_at_point_agrees_with_m31_embedding(oracle):
    rnd = np.random.default_rng(11)
    n = 6
    coeffs = rnd.integers(0, P, 1 << n, dtype=np.uint32)
    L = oracle.lib()
    xy = np.zeros(2, np.uint32)
    L.orc_circle_point(987654321, O.ptr(xy))
    pt = O.u32([xy[0], 0, 0, 0, xy[1], 0, 0, 0])
    out = oracle.eval_at_point(coeffs, pt)
    assert int(out[0]) == L.orc_eval_basis_at_m31_point(O.ptr(coeffs), n, int(xy[0]), int(xy[1]))
    assert (out[1:] == 0).all()


def test_twiddle_layout(oracle):
    """K2: layer i = x-coords of the first half of coset.double^i, bit-reversed; trailing 1."""
    h = 5
    tw, itw = oracle.Twiddles(h).arrays()
    assert tw[-1] == 1
    L = oracle.lib()
    off = 0
    init, step = 1 << (31 - h - 2), 1 << (31 - h)
    for layer in range(h):
        size = 1 << (h - layer)
        xs = [pt_mul(GEN, (init + k * step) % (1 << 31))[0] for k in range(size // 2)]
        lg = h - layer - 1
        for k in range(size // 2):
            assert tw[off + k] == xs[L.orc_bit_reverse_index(k, lg)]
        off += size // 2
        init, step = (2 * init) % (1 << 31), (2 * step) % (1 << 31)
    assert all(int(a) * int(b) % P == 1 for a, b in zip(tw, itw))


def _batches_single_point(oracle, polys, n, seed):
    """sample every column at one random secure point; returns flat batch description."""
    L = oracle.lib()
    ch = C.c_void_p(L.orc_channel_new())
    L.orc_channel_mix_u64(ch, seed)
    pt = np.zeros(8, np.uint32)
    L.orc_get_random_point(ch, O.ptr(pt))
    alpha = np.zeros(4, np.uint32)
    L.orc_channel_draw_secure_felt(ch, O.ptr(alpha))
    L.orc_channel_free(ch)
    vals = np.concatenate([oracle.eval_at_point(p, pt) for p in polys])
    return pt, alpha, vals


def test_deep_quotient_is_low_degree(oracle):
    """the DEEP quotient of blow-up-2 LDEs is low degree in every coordinate."""
    rnd = np.random.default_rng(12)
    n, ncols = 5, 3
    tw = oracle.Twiddles(n + 1)
    polys = [rnd.integers(0, P, 1 << n, dtype=np.uint32) for _ in range(ncols)]
    ldes = [tw.evaluate(p, n + 1) for p in polys]
    pt, alpha, vals = _batches_single_point(oracle, polys, n, 99)
    outs = [np.zeros(1 << (n + 1), np.uint32) for _ in range(4)]
    counts = np.array([ncols], np.int32)
    idx = np.arange(ncols, dtype=np.int32)
    oracle.lib().orc_accumulate_quotients(n + 1, O.ptr_array(ldes), ncols, O.ptr(alpha), 1, O.ptr(pt), O.ptr(counts), O.ptr(idx),
                                          O.ptr(vals), 1, O.ptr_array(outs))
    for q in outs:
        co = tw.interpolate(q)
        assert (co[1 << n:] == 0).all() and co[:1 << n].any()
    # a wrong sampled value breaks low-degreeness
    vals2 = vals.copy()
    vals2[0] ^= 1
    oracle.lib().orc_accumulate_quotients(n + 1, O.ptr_array(ldes), ncols, O.ptr(alpha), 1, O.ptr(pt), O.ptr(counts), O.ptr(idx),
                                          O.ptr(vals2), 1, O.ptr_array(outs))
    assert any(tw.interpolate(q)[1 << n:].any() for q in outs)


def test_fri_folds_halve_the_degree(oracle):
    """fold_circle_into_line / fold_line of a low-degree evaluation stay low degree:
    after folding a degree-<2^n circle poly (blow-up 2) log-many times the last layer is constant."""
    rnd = np.random.default_rng(13)
    n = 5
    L = oracle.lib()
    tw = oracle.Twiddles(n + 1)
    src = [tw.evaluate(rnd.integers(0, P, 1 << n, dtype=np.uint32), n + 1) for _ in range(4)]
    alpha = O.u32([3, 1, 4, 1])
    dst = [np.zeros(1 << n, np.uint32) for _ in range(4)]
    L.orc_fold_circle_into_line(O.ptr_array(dst), O.ptr_array(src), n + 1, O.ptr(alpha))
    log, dbl = n, 0
    cur = dst
    while log > 1:
        nxt = [np.zeros(1 << (log - 1), np.uint32) for _ in range(4)]
        L.orc_fold_line_dom(O.ptr_array(cur), log, dbl, O.ptr(alpha), O.ptr_array(nxt))
        cur, log, dbl = nxt, log - 1, dbl + 1
    # degree bound 2^n/… -> after n-1 line folds of a degree < 2^(n-1) line poly: size-2 domain, constant
    for c in cur:
        assert c[0] == c[1]
    # a high-degree input does not fold to a constant
    bad = [rnd.integers(0, P, 1 << (n + 1), dtype=np.uint32) for _ in range(4)]
    dst = [np.zeros(1 << n, np.uint32) for _ in range(4)]
    L.orc_fold_circle_into_line(O.ptr_array(dst), O.ptr_array(bad), n + 1, O.ptr(alpha))
    log, dbl, cur = n, 0, dst
    while log > 1:
        nxt = [np.zeros(1 << (log - 1), np.uint32) for _ in range(4)]
        L.orc_fold_line_dom(O.ptr_array(cur), log, dbl, O.ptr(alpha), O.ptr_array(nxt))
        cur, log, dbl = nxt, log - 1, dbl + 1
    assert any(c[0] != c[1] for c in cur)


def test_synthetic_trace_satisfies_air(oracle):
    comp = np.array([6, 3, 37, 20], np.int32)
    w = 3 + 37 + 20
    rows = np.zeros((64, w), np.uint32)
    oracle.lib().orc_synth_rows(O.ptr(comp), 0, 5, 9, 0, 64, O.ptr(rows))
    pre, main, inter = rows[:, :3].astype(object), rows[:, 3:40].astype(object), rows[:, 40:].astype(object)
    assert pre[0, 0] == 1 and pre[1:, 0].sum() == 0 and pre[63, 1] == 1 and pre[:63, 1].sum() == 0
    for r in range(63):
        assert (main[r + 1, 0] - main[r, 0] - 1) % P == 0
        assert (main[r + 1, 1] - main[r, 1] - main[r, 0]) % P == 0
    for k in range(2, 37):
        if k % 16 >= 2:
            assert all((main[r, k] - main[r, k - 1] ** 2 - main[r, k - 2] ** 2) % P == 0 for r in range(64))
    for k in range(20):
        if k % 16 >= 2:
            assert all((inter[r, k] - inter[r, k - 1] ** 2 - inter[r, k - 2] ** 2) % P == 0 for r in range(64))
    assert rows.max() < P


@pytest.mark.parametrize("hash_mode", [O.HASH_STD, O.HASH_RAW0])
@pytest.mark.parametrize("fri_mode", [O.FRI_ALPHA_PREV, O.FRI_ALPHA_FIRST])
@pytest.mark.parametrize("lcd", [1, 2])
def test_prove_verify_roundtrip_and_tamper(oracle, hash_mode, fri_mode, lcd):
    """Mirrors the reference's prove->verify round trips (prover/src/machine.rs:505-533) on the
    synthetic AIR, plus tampering (which the reference never tests)."""
    cfg = O.default_cfg(pow_bits=6, hash_mode=hash_mode, fri_alpha_mode=fri_mode, log_constraint_degree=lcd)
    comps = [(7, 3, 21, 6), (5, 2, 4, 3), (7, 2, 3, 0)]
    w = oracle.prove_synth(comps, cfg, seed=3, ad=b"\x01\x02")
    assert oracle.verify_synth(comps, cfg, w, ad=b"\x01\x02") is None
    assert oracle.verify_synth(comps, cfg, w, ad=b"\x01\x03") is not None       # transcript prefix matters
    assert oracle.verify_synth([(7, 3, 21, 6), (5, 2, 4, 3), (7, 2, 4, 0)], cfg, w, ad=b"\x01\x02") is not None
    rnd = random.Random(hash_mode * 4 + fri_mode * 2 + lcd)
    # every word of the proof is bound: flip one bit anywhere -> reject
    for pos in [rnd.randrange(6, len(w)) for _ in range(60)] + [len(w) - 1, len(w) - 4]:
        w2 = w.copy()
        w2[pos] ^= 1 << rnd.randrange(31)
        assert oracle.verify_synth(comps, cfg, w2, ad=b"\x01\x02") is not None, pos
    # determinism
    assert (oracle.prove_synth(comps, cfg, seed=3, ad=b"\x01\x02", threads=1) == w).all()


def test_proof_differs_between_switchable_rules(oracle):
    comps = [(6, 2, 5, 2)]
    ws = {}
    for hm in (0, 1):
        for fm in (0, 1):
            ws[(hm, fm)] = oracle.prove_synth(comps, O.default_cfg(pow_bits=4, hash_mode=hm, fri_alpha_mode=fm))
    assert len({w.tobytes() for w in ws.values()}) == 4


def test_golden_fixtures(oracle):
    """Committed fixtures (tests/golden/oracle_golden.json, made by tests/golden/make_golden.py from this oracle
    after the independent checks above passed) freeze the oracle so later edits cannot drift silently."""
    path = os.path.join(GOLDEN, "oracle_golden.json")
    g = json.load(open(path))
    for case in g["prove"]:
        cfg = O.default_cfg(**case["cfg"])
        w = oracle.prove_synth([tuple(c) for c in case["comps"]], cfg, seed=case["seed"], ad=bytes(case["ad"]))
        assert hashlib.sha256(w.tobytes()).hexdigest() == case["sha256"], case
        assert [int(x) for x in w[6:14]] == case["root0"]
    for case in g["lde_commit"]:
        rnd = np.random.default_rng(case["seed"])
        cols = [rnd.integers(0, P, 1 << l, dtype=np.uint32) for l in case["logs"]]
        tw = oracle.Twiddles(max(case["logs"]))
        ldes = [tw.evaluate(tw.interpolate(c), int(np.log2(len(c))) + 1) for c in cols]
        for mode, key in ((O.HASH_STD, "root_std"), (O.HASH_RAW0, "root_raw0")):
            assert [int(x) for x in oracle.merkle_commit(ldes, mode)] == case[key]


# ---------------- oracle/backend_ops.h: pinned by the identities of the construction ----------------

def test_oracle_batch_inverse_is_the_inverse(oracle):
    import ctypes as C
    L = oracle.lib()
    rng = np.random.default_rng(1234)
    v = rng.integers(1, O.P, 1000, dtype=np.uint32)
    inv = np.zeros_like(v)
    L.orc_batch_inverse_m31(O.ptr(v), O.ptr(inv), C.c_size_t(len(v)))
    assert all((int(a) * int(b)) % O.P == 1 for a, b in zip(v, inv))
    q = [rng.integers(1, O.P, 64, dtype=np.uint32) for _ in range(4)]
    qi = [np.zeros(64, np.uint32) for _ in range(4)]
    L.orc_batch_inverse_qm31(O.ptr_array(q), O.ptr_array(qi), C.c_size_t(64))
    for i in range(64):
        a = np.array([c[i] for c in q], np.uint32); b = np.array([c[i] for c in qi], np.uint32)
        out = np.zeros(4, np.uint32)
        L.orc_qm31_mul(O.ptr(a), O.ptr(b), O.ptr(out))
        assert list(out) == [1, 0, 0, 0]


def test_oracle_fri_decompose_splits_off_the_sign_component(oracle):
    """f = g + lambda * v_n with v_n = +1 / -1 on the halves, and g has no v_n component (its own decomposition coefficient is 0)."""
    L = oracle.lib()
    log = 6
    rng = np.random.default_rng(5)
    src = [rng.integers(0, O.P, 1 << log, dtype=np.uint32) for _ in range(4)]
    g = [np.zeros(1 << log, np.uint32) for _ in range(4)]
    lam = np.zeros(4, np.uint32)
    L.orc_fri_decompose(O.ptr_array(src), log, O.ptr_array(g), O.ptr(lam))
    half = 1 << (log - 1)
    for q in range(4):
        assert np.array_equal((g[q][:half].astype(np.uint64) + lam[q]) % O.P, src[q][:half])
        assert np.array_equal((g[q][half:].astype(np.uint64) + O.P - lam[q]) % O.P, src[q][half:])
    g2 = [np.zeros(1 << log, np.uint32) for _ in range(4)]
    lam2 = np.ones(4, np.uint32)
    L.orc_fri_decompose(O.ptr_array(g), log, O.ptr_array(g2), O.ptr(lam2))
    assert not lam2.any()


def test_oracle_commit_on_layer_rebuilds_the_committed_tree(oracle):
    import ctypes as C
    L = oracle.lib()
    rng = np.random.default_rng(9)
    big = [rng.integers(0, O.P, 1 << 5, dtype=np.uint32) for _ in range(18)]
    small = [rng.integers(0, O.P, 1 << 3, dtype=np.uint32) for _ in range(2)]
    for mode in (0, 1):
        root, layers = oracle.merkle_commit(big + small, mode, want_layers=True)
        prev, off = None, 0
        for log in range(5, -1, -1):
            cols = big if log == 5 else small if log == 3 else []
            out = np.zeros(8 << log, np.uint32)
            L.orc_commit_on_layer(log, O.ptr(prev) if prev is not None else None, O.ptr_array(cols) if cols else None, C.c_size_t(len(cols)), mode, O.ptr(out))
            assert np.array_equal(out, layers[off:off + (8 << log)]), (mode, log)
            off += 8 << log
            prev = out
        assert np.array_equal(prev, root)
