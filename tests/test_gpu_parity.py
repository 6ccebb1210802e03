"""GPU parity tests (-m gpu): every C-ABI entry of libnexus_hip.so against the CPU oracle, bit-exact,
on identical seeded inputs; then size-independent properties at BASELINE sizes.

Mirrors the reference's test strategy: per-op checks in the style of
prover/src/test_utils.rs::commit_traces (real commit path), the layout pin `test_order`
(prover/src/trace/utils.rs:117-128) and prove->verify round trips (prover/src/machine.rs:505-533).
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest
import torch  # noqa: F401  (before libnexus_hip.so: the wheel bundles its own ROCm runtime and whichever HIP runtime is loaded first serves
#                 the whole process; loading /opt/rocm's first leaves torch without a device — only the multi-rank transport tests use torch)

import oracle_lib as O

pytestmark = pytest.mark.gpu
P = O.P
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(autouse=True)
def _small_sharded_fri_layers(monkeypatch):
    """Row-sharded proves keep a FRI layer sharded only from 2^21 rows up (NX_FRI_DIST_MIN_LOG): the tests lower the limit so that the
    sharded layer trees and folds run at test sizes too."""
    monkeypatch.setenv("NX_FRI_DIST_MIN_LOG", "0")


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


def rand_cols(seed, n_cols, log):
    return np.random.default_rng(seed).integers(0, P, (n_cols, 1 << log), dtype=np.uint32)


def test_twiddles(be, oracle):
    for h in (1, 2, 5, 11, 16):
        tw, itw = be.precompute_twiddles(h).to_cpu()
        otw, oitw = oracle.Twiddles(h).arrays()
        assert np.array_equal(tw, otw) and np.array_equal(itw, oitw), h


@pytest.mark.parametrize("log", [1, 2, 3, 4, 6, 9, 12, 13, 14, 16, 18])
def test_interpolate_evaluate_match_oracle(be, oracle, log):
    n_cols = 3 if log > 14 else 19
    tw = be.precompute_twiddles(log + 1)
    otw = oracle.Twiddles(log + 1)
    vals = rand_cols(log, n_cols, log)
    cols = be.columns_from_host(vals)
    be.interpolate_columns(tw, cols)
    coeffs = cols.to_cpu()
    ref = np.stack([otw.interpolate(v) for v in vals])
    assert np.array_equal(coeffs, ref)
    for expand in (0, 1, 2):
        lde = be.evaluate_polynomials(tw, cols, expand)   # domain log+expand <= log+2 is covered by the tree
        got = lde.to_cpu()
        for c in range(min(n_cols, 4)):
            assert np.array_equal(got[c], otw.evaluate(ref[c], log + expand)), (log, expand, c)
        assert got.max() < P
        lde.free()


@pytest.mark.parametrize("log", [13, 14, 15, 16, 17, 18, 19, 20, 21, 22])
def test_fused_lde_matches_oracle_at_every_pass_shape(be, oracle, log):
    """nx_lde_batch with blow-up 2 takes the fused-middle route from 2^14 rows up (lde_mid_kernel with 1..7 layers here, 8 and 9 in
    the large-transform test, the 3-pass plan at 2^23): coefficients and every LDE value against the oracle, an odd column count so
    the batch tail is exercised too."""
    n_cols = 3
    vals = rand_cols(700 + log, n_cols, log)
    tw = be.precompute_twiddles(log)
    otw = oracle.Twiddles(log + 1)
    cols = be.columns_from_host(vals)
    lde = be.lde(tw, cols, 1)
    coeffs = np.stack([otw.interpolate(v) for v in vals])
    assert np.array_equal(cols.to_cpu(), coeffs)
    got = lde.to_cpu()
    for c in range(n_cols):
        assert np.array_equal(got[c], otw.evaluate(coeffs[c], log + 1)), (log, c)


def test_context_options_are_per_context_and_validated(nz):
    a, b = nz.HipBackend(0), nz.HipBackend(0)
    try:
        a.set_option("fft.streams", 1); b.set_option("fft.streams", 3)
        a.set_option("dist.chunks", 2)
        assert a.get_option("fft.streams") == 1 and b.get_option("fft.streams") == 3
        assert a.get_option("dist.chunks") == 2 and b.get_option("dist.chunks") == 0
        for name, v in (("fft.streams", 9), ("air.quarter_domain", 3), ("fft.pipe", 1), ("no.such.option", 1), ("air.segment", 1)):
            with pytest.raises(nz.NexusHipError):
                a.set_option(name, v)
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("log", [21, 22, 23, 24])
def test_large_transforms_match_oracle_and_roundtrip(be, oracle, log):
    """2-pass schedules of the wide kernel — 2^13-row tiles (13+8, 13+9 layers) up to 2^22 points, 2^14-row tiles (14+9, 14+10, 14+11)
    from 2^23 points (round 4: the third pass is gone) —: one column against the oracle, the others through interpolate∘evaluate = id
    and LDE-restricted-to-the-trace-domain properties.  log 23 / 24 run interpolate, evaluate and the inverse of the extension at
    2^23 ... 2^25 points, i.e. every 2^14-tile kernel (FIRST and K = 9, 10, 11, forward and inverse)."""
    n_cols = 3
    tw = be.precompute_twiddles(log + 1)
    otw = oracle.Twiddles(log + 1)
    vals = rand_cols(900 + log, n_cols, log)
    cols = be.columns_from_host(vals)
    be.interpolate_columns(tw, cols)
    coeffs = cols.to_cpu()
    assert np.array_equal(coeffs[2], otw.interpolate(vals[2]))
    back = be.evaluate_polynomials(tw, cols, 0)
    assert np.array_equal(back.to_cpu(), vals)
    back.free()
    lde = be.evaluate_polynomials(tw, cols, 1)
    got = lde.to_cpu()
    assert got.max() < P
    assert np.array_equal(got[1], otw.evaluate(coeffs[1], log + 1))
    be.interpolate_columns(tw, lde)
    c2 = lde.to_cpu()
    assert np.array_equal(c2[:, :1 << log], coeffs) and not c2[:, 1 << log:].any()


def test_lde_fused_and_pointer_table_path(be, oracle):
    """nx_lde_batch == interpolate+evaluate; columns that are not a uniform slab take the table path."""
    log, n_cols = 10, 7
    tw = be.precompute_twiddles(log)
    otw = oracle.Twiddles(log)
    vals = rand_cols(77, n_cols, log)
    cols = be.columns_from_host(vals)
    lde = be.lde(tw, cols, 1)
    ref_c = np.stack([otw.interpolate(v) for v in vals])
    assert np.array_equal(cols.to_cpu(), ref_c)
    assert np.array_equal(lde.to_cpu(), np.stack([otw.evaluate(c, log + 1) for c in ref_c]))
    # scattered columns: reversed pointer order -> not a constant positive stride
    cols2 = be.columns_from_host(vals)
    ptrs = (C.c_void_p * n_cols)(*[cols2.ptr.value + (n_cols - 1 - i) * (4 << log) for i in range(n_cols)])
    be._chk(be.L.nx_interpolate_batch(be.ctx, tw.h, ptrs, n_cols, log))
    assert np.array_equal(cols2.to_cpu(), ref_c)


def test_layout_permutations(be, oracle):
    for log in (1, 3, 8, 13):
        nat = rand_cols(log + 100, 3, log)
        d = be.columns_from_host(nat)
        fin = be.finalize_columns(d).to_cpu()
        for c in range(3):
            assert np.array_equal(fin[c], oracle.finalize_column(nat[c]))
        up = be.upload_coset_order(nat[0]).to_cpu()[0]
        assert np.array_equal(up, fin[0])
        # whole-trace upload: pinned in place, streamed, permuted on device
        up_all = be.upload_columns(list(nat)).to_cpu()
        assert np.array_equal(up_all, fin)
        assert np.array_equal(be.upload_columns(list(nat), coset_order=False).to_cpu(), nat)
        # K1: bit_reverse_column
        br = be.columns_from_host(nat[:1])
        be.bit_reverse_column(br)
        ref = nat[0].copy()
        oracle.lib().orc_bit_reverse(O.ptr(ref), log)
        assert np.array_equal(br.to_cpu()[0], ref)
    # the reference's own test_order (prover/src/trace/utils.rs:117-128) at log 3
    vals = np.arange(8, dtype=np.uint32)
    col = be.finalize_columns(be.columns_from_host(vals)).to_cpu()[0]
    L = oracle.lib()
    for i in range(8):
        assert col[i] == vals[L.orc_bit_reverse_index(L.orc_coset_index_to_circle_domain_index(i, 3), 3)]


@pytest.mark.parametrize("mode", [0, 1])
def test_merkle_commit_matches_oracle(be, oracle, mode):
    be.set_hash_mode(mode)
    try:
        for logs in ([6] * 5, [6] * 16, [6] * 17, [9] * 40, [7, 4, 7, 5, 4, 7, 0], [3], []):
            host = [np.random.default_rng(len(logs) * 31 + i).integers(0, P, 1 << l, dtype=np.uint32) for i, l in enumerate(logs)]
            sets = [be.columns_from_host(h) for h in host]
            tree = be.merkle_commit(sets)
            root, layers = oracle.merkle_commit(host, mode, want_layers=True)
            assert np.array_equal(tree.root(), root), (mode, logs)
            mx = max(logs) if logs else 0
            off = 0
            for k in range(mx, -1, -1):
                assert np.array_equal(tree.layer(k).reshape(-1), layers[off:off + (8 << k)]), (mode, logs, k)
                off += 8 << k
    finally:
        be.set_hash_mode(0)


def test_eval_at_points_matches_oracle(be, oracle):
    L = oracle.lib()
    for log in (0, 1, 5, 11, 12, 15):
        polys = rand_cols(log + 7, 5, log)
        d = be.columns_from_host(polys)
        ch = C.c_void_p(L.orc_channel_new())
        L.orc_channel_mix_u64(ch, log)
        p1, p2 = np.zeros(8, np.uint32), np.zeros(8, np.uint32)
        L.orc_get_random_point(ch, O.ptr(p1))
        L.orc_get_random_point(ch, O.ptr(p2))
        L.orc_channel_free(ch)
        idx = [0, 1, 2, 3, 4, 0, 3]
        pts = [p1, p1, p1, p1, p1, p2, p2]
        got = be.eval_at_points(d, idx, pts)
        for i, (pi, pt) in enumerate(zip(idx, pts)):
            assert np.array_equal(got[i], oracle.eval_at_point(polys[pi], pt)), (log, i)


def _random_point_and_alpha(oracle, seed):
    L = oracle.lib()
    ch = C.c_void_p(L.orc_channel_new())
    L.orc_channel_mix_u64(ch, seed)
    p1, p2, a = np.zeros(8, np.uint32), np.zeros(8, np.uint32), np.zeros(4, np.uint32)
    L.orc_get_random_point(ch, O.ptr(p1))
    L.orc_get_random_point(ch, O.ptr(p2))
    L.orc_channel_draw_secure_felt(ch, O.ptr(a))
    L.orc_channel_free(ch)
    return p1, p2, a


def test_accumulate_quotients_matches_oracle(be, oracle):
    for log, n_cols in ((4, 3), (9, 21), (12, 40)):
        cols = rand_cols(log * 13, n_cols, log)
        p1, p2, alpha = _random_point_and_alpha(oracle, log)
        rng = np.random.default_rng(log)
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
                                              O.ptr(counts), O.ptr(cidx), O.ptr(vals), 4, O.ptr_array(outs))
        assert np.array_equal(got, np.stack(outs)), log


def test_fri_folds_match_oracle(be, oracle):
    L = oracle.lib()
    for log in (2, 3, 6, 12):
        tw = be.precompute_twiddles(max(log, 1))
        src = rand_cols(log + 50, 4, log)
        dst0 = rand_cols(log + 51, 4, log - 1)
        _, _, alpha = _random_point_and_alpha(oracle, log + 9)
        d_src, d_dst = be.columns_from_host(src), be.columns_from_host(dst0)
        be.fold_circle_into_line(tw, d_dst, d_src, alpha)
        ref = [np.ascontiguousarray(c).copy() for c in dst0]
        L.orc_fold_circle_into_line(O.ptr_array(ref), O.ptr_array([np.ascontiguousarray(c) for c in src]), log, O.ptr(alpha))
        assert np.array_equal(d_dst.to_cpu(), np.stack(ref)), ("circle", log)
        for dbl in (0, 2):
            out = be.fold_line(tw, d_src, alpha, dbl).to_cpu()
            ref2 = [np.zeros(1 << (log - 1), np.uint32) for _ in range(4)]
            L.orc_fold_line_dom(O.ptr_array([np.ascontiguousarray(c) for c in src]), log, dbl, O.ptr(alpha), O.ptr_array(ref2))
            assert np.array_equal(out, np.stack(ref2)), ("line", log, dbl)


def test_grind_matches_oracle(be, oracle):
    L = oracle.lib()
    for seed, bits in ((1, 0), (2, 6), (3, 12), (4, 17)):
        ch = C.c_void_p(L.orc_channel_new())
        L.orc_channel_mix_u64(ch, seed)
        d = np.zeros(8, np.uint32)
        L.orc_channel_digest(ch, O.ptr(d))
        assert be.grind(d, bits) == L.orc_channel_grind(ch, bits)
        L.orc_channel_free(ch)


def test_synth_trace_fill_matches_oracle(be, oracle):
    comps = [(9, 5, 37, 20), (5, 2, 3, 0)]
    for tree in range(3):
        got = be.synth_fill_tree(comps, tree, seed=11, inter_seed=0xABCDEF0123)
        ref = oracle.synth_tree_columns(comps, tree, 11, 0xABCDEF0123)
        flat = [row for s in got for row in s.to_cpu()]
        assert len(flat) == len(ref)
        for a, b in zip(flat, ref):
            assert np.array_equal(a, b), tree


PROVE_CASES = [
    ([(8, 3, 20, 6)], dict(pow_bits=8)),
    ([(10, 27, 40, 8), (6, 2, 5, 4)], dict(pow_bits=10)),
    ([(9, 4, 18, 4)], dict(pow_bits=5, log_constraint_degree=2)),
    ([(8, 3, 20, 6), (8, 2, 3, 0), (5, 2, 2, 2)], dict(pow_bits=6, hash_mode=1, fri_alpha_mode=1)),
    ([(12, 27, 347, 64)], dict(pow_bits=10)),
    ([(13, 4, 33, 0), (11, 3, 17, 16), (4, 2, 2, 0)], dict(pow_bits=7, log_constraint_degree=2)),
    ([(15, 8, 61, 20), (14, 3, 17, 16)], dict(pow_bits=9)),                       # every transform through the multi-pass wide kernel (13 + 2/3 layers)
    ([(16, 5, 35, 0)], dict(pow_bits=8, log_constraint_degree=2)),                # constraint domain 2^18: replicas x4 in the LDE top pass
    # per-component constraint-degree bounds (reference: v1 main +2, extensions +1; prover2 1 / shifts 2): the composition polynomial
    # takes the largest log_size + bound — here 2^(12+1), not the 2^(12+2) a global bound of 2 gives
    ([(12, 4, 33, 8, 1), (9, 3, 17, 16, 2), (5, 2, 2, 0, 1)], dict(pow_bits=7, log_constraint_degree=2)),
    ([(11, 5, 21, 4, 2), (11, 2, 6, 0, 1), (8, 2, 3, 4)], dict(pow_bits=6, log_constraint_degree=2)),   # main +2, extensions +1 / defaulted
]


@pytest.mark.parametrize("comps,kw", PROVE_CASES)
def test_prove_bit_exact_vs_oracle(be, nz, oracle, comps, kw):
    cfg = nz.default_config(**kw)
    ocfg = O.default_cfg(**kw)
    ad = b"\x07\x01"
    words = be.prove(comps, cfg, seed=0xC0FFEE, ad=ad)
    assert oracle.verify_synth(comps, ocfg, words, ad=ad) is None
    ref = oracle.prove_synth(comps, ocfg, seed=0xC0FFEE, ad=ad, threads=8)
    assert len(ref) == len(words)
    if not np.array_equal(ref, words):
        bad = int(np.nonzero(ref != words)[0][0])
        pytest.fail(f"first differing proof word {bad} of {len(ref)} (roots are words 6..37)")


def test_golden_fixtures_through_hip(be, nz):
    g = json.load(open(os.path.join(GOLDEN, "oracle_golden.json")))
    for case in g["prove"]:
        w = be.prove([tuple(c) for c in case["comps"]], nz.default_config(**case["cfg"]), seed=case["seed"], ad=bytes(case["ad"]))
        assert hashlib.sha256(w.tobytes()).hexdigest() == case["sha256"], case["comps"]
    for case in g["lde_commit"]:
        rnd = np.random.default_rng(case["seed"])
        cols = [rnd.integers(0, P, 1 << l, dtype=np.uint32) for l in case["logs"]]
        tw = be.precompute_twiddles(max(case["logs"]))
        ldes = []
        for c in cols:
            d = be.columns_from_host(c)
            ldes.append(be.lde(tw, d, 1))
        for mode, key in ((0, "root_std"), (1, "root_raw0")):
            be.set_hash_mode(mode)
            assert [int(x) for x in be.merkle_commit(ldes).root()] == case[key]
        be.set_hash_mode(0)


# ---------------- BASELINE-size properties (size-independent checks; the oracle would take minutes) --------

def test_config2_shape_lde_roundtrip_and_root_consistency(be):
    """BASELINE config #2 shape (2^20 rows) on a column subset: interpolate∘evaluate = id, the LDE restricted
    to blow-up 0 reproduces the input, and the Merkle root is identical through the slab and the
    pointer-table path."""
    log, n_cols = 20, 24
    vals = rand_cols(2020, n_cols, log)
    tw = be.precompute_twiddles(log)
    cols = be.columns_from_host(vals)
    lde = be.lde(tw, cols, 1)                      # cols now holds coefficients
    back = be.evaluate_polynomials(tw, cols, 0)    # evaluate on the original domain
    assert np.array_equal(back.to_cpu(), vals)
    l = lde.to_cpu()
    assert l.max() < P
    # interpolating the LDE gives the same coefficients, zero-extended
    be.interpolate_columns(tw, lde)
    c2 = lde.to_cpu()
    assert np.array_equal(c2[:, :1 << log], cols.to_cpu()) and not c2[:, 1 << log:].any()
    t1 = be.merkle_commit([back]).root()
    singles = [be.columns_from_host(v) for v in vals]
    t2 = be.merkle_commit(singles).root()
    assert np.array_equal(t1, t2)


def test_config2_lde_and_commit_bit_exact_vs_oracle_at_full_height(be, oracle):
    """BASELINE config #2 at its full height (2^20 rows -> 2^21 LDE rows) on a column subset the oracle finishes in seconds: every
    LDE value, every coefficient and the Blake2s Merkle root of nx_lde_commit equal the CPU oracle's."""
    log, n_cols = 20, 12
    vals = rand_cols(2021, n_cols, log)
    tw = be.precompute_twiddles(log)
    cols = be.columns_from_host(vals)
    lde, root = be.lde_commit(tw, cols, 1)
    otw = oracle.Twiddles(log + 1)
    coeffs = [otw.interpolate(v) for v in vals]
    ext = [otw.evaluate(c, log + 1) for c in coeffs]
    assert np.array_equal(cols.to_cpu(), np.stack(coeffs))
    assert np.array_equal(lde.to_cpu(), np.stack(ext))
    assert np.array_equal(root, oracle.merkle_commit(ext))


def test_large_prove_accepted_by_oracle_verifier(be, nz, oracle):
    """A 2^18-row synthetic prove (347 main / 27 preprocessed / 64 interaction columns): the oracle verifier
    must accept; tampering must be rejected."""
    comps = [(18, 27, 347, 64)]
    cfg, ocfg = nz.default_config(), O.default_cfg()
    w, stats = be.prove(comps, cfg, seed=22, want_stats=True)
    assert oracle.verify_synth(comps, ocfg, w) is None
    w2 = w.copy()
    w2[len(w) // 2] ^= 4
    assert oracle.verify_synth(comps, ocfg, w2) is not None
    assert stats["total"] > 0 and stats["lde_kernel_ms"] > 0


@pytest.mark.parametrize("world,n_cols,log,mode", [(2, 21, 10, 0), (4, 70, 13, 0), (8, 33, 12, 1)])
def test_transposed_commit_on_device_matches_single_commit(nz, world, n_cols, log, mode):
    """The transposed commit (sharded.py) with HipShardOps: W ranks as threads, one context each on this GPU, tensors handed over in
    process; every rank's root equals nx_merkle_commit of all the LDE columns on one context."""
    import threading, torch
    from nexus_zkvm_amd.sharded import HipShardOps, transposed_commit
    dev = torch.device("cuda:0")
    cols = np.random.default_rng(99).integers(0, P, (n_cols, 1 << log), dtype=np.uint32)
    cuts = [n_cols * r // world for r in range(world + 1)]
    if world > 2:
        cuts[2] = cuts[1]                                              # rank 1 holds no columns
    ranges = [(cuts[r], cuts[r + 1]) for r in range(world)]
    ref_be = nz.HipBackend(0); ref_be.set_hash_mode(mode)
    tw = ref_be.precompute_twiddles(log + 1)
    full = ref_be.lde(tw, ref_be.columns_from_host(cols), 1)
    expected = ref_be.merkle_commit([full]).root()
    bar = threading.Barrier(world)
    mail = {}
    roots, errs = [None] * world, []

    class Comm:
        def __init__(self, rank):
            self.rank, self.world, self.torch, self.device = rank, world, torch, dev
        def empty_state(self, n_rows):
            return torch.empty((n_rows, 8), dtype=torch.int32, device=dev)
        def all_gather(self, out_list, t):
            mail[("ag", self.rank)] = t
            bar.wait()
            for q in range(world):
                out_list[q].copy_(mail[("ag", q)])
            torch.cuda.synchronize(); bar.wait()
        def device_sync(self):
            torch.cuda.synchronize()

    def exchange(comm, send):
        for s_, t in enumerate(send):
            mail[("a2a", comm.rank, s_)] = t
        torch.cuda.synchronize(); bar.wait()
        recv = [mail[("a2a", q, comm.rank)] for q in range(world)]
        recv = [t.clone() if t is not None else None for t in recv]
        torch.cuda.synchronize(); bar.wait()
        return recv

    def worker(rank):
        try:
            b = nz.HipBackend(0); b.set_hash_mode(mode)
            ops = HipShardOps(b, b.precompute_twiddles(log + 1))
            lo, hi = ranges[rank]
            local = ops.lde(b.columns_from_host(cols[lo:hi]), 1) if hi > lo else b.columns(0, log + 1)
            roots[rank], _ = transposed_commit(ops, Comm(rank), local, ranges, log + 1, exchange=exchange)
        except Exception as e:                                          # surface the failure instead of dead-locking the barrier
            errs.append(e); bar.abort()
    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    for r in range(world):
        assert np.array_equal(roots[r], expected), r


def test_headline_prove_verifies_and_the_session_reproduces_it(be, nz, oracle):
    """BASELINE config #3 at full size (2^22 rows, 27 + 347 + 64 columns): the proof the bench times is accepted by the oracle's
    verifier (a check whose cost does not grow with the trace), a tampered one is not, and the generic session — the same machine as
    a recorded AIR, traces filled on the device into session-owned columns — emits the same bytes."""
    import ctypes as C
    import nexus_zkvm_amd.air_program as ap
    from test_air_program_cpu import synthetic_program
    log, n_pre, n_main, n_inter = 22, 27, 347, 64
    comps = [(log, n_pre, n_main, n_inter)]
    cfg, ocfg = nz.default_config(), O.default_cfg()
    w = be.prove(comps, cfg, seed=1)
    assert oracle.verify_synth(comps, ocfg, w) is None
    w2 = w.copy(); w2[len(w) // 3] ^= 1
    assert oracle.verify_synth(comps, ocfg, w2) is not None
    cols = [(0, k) for k in range(n_pre)] + [(1, k) for k in range(n_main)] + [(2, k) for k in range(n_inter)]
    comp = ap.Component(log, synthetic_program(ap, n_pre, n_main, n_inter), cols)
    carr = be._comps(comps)
    s = be.prover_session(cfg, log)

    def fill(tree, n, inter_seed=0):
        ptrs = s.tree_begin([log] * n)
        be._chk(be.L.nx_synth_fill_tree(be.ctx, carr, 1, tree, C.c_uint64(1), C.c_uint64(inter_seed), (C.c_void_p * n)(*ptrs)))
        return s.tree_commit()
    s.mix_u64(log)
    fill(0, n_pre); fill(1, n_main)
    z = s.draw_felt()
    s.mix_felts(np.zeros(4, np.uint32))
    fill(2, n_inter, (int(z[0]) << 32) ^ int(z[1]) ^ (int(z[2]) << 16) ^ (int(z[3]) << 48))
    assert np.array_equal(s.prove([comp]), w)
    s.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_sharded_leaf_chain_equals_single_commit(be, mode):
    """SURVEY §8(e) / config #4 building blocks on one GPU: the columns of a tree are cut into 16-aligned shards (as
    nexus_zkvm_amd.sharded.plan_column_shards would for 3 ranks), each shard continues the chaining state of the
    previous one over row chunks, the last one finalises; leaves and root must equal nx_merkle_commit of all columns."""
    from nexus_zkvm_amd.sharded import plan_column_shards
    be.set_hash_mode(mode)
    try:
        for n_cols, log in ((40, 9), (33, 7), (16, 6), (347, 8)):
            vals = rand_cols(n_cols * 7 + log, n_cols, log)
            full = be.columns_from_host(vals)
            tree = be.merkle_commit([full])
            shards = [s for s in plan_column_shards(n_cols, 3) if s[1] > s[0]]
            n_rows = 1 << log
            state = be.columns(8, log)            # n_rows x 8 words
            bounds = [0, n_rows // 3, n_rows]
            for si, (lo, hi) in enumerate(shards):
                part = be.columns_from_host(vals[lo:hi])
                for rb, re in zip(bounds, bounds[1:]):
                    sp = state.ptr.value + rb * 32
                    be.merkle_leaf_chain(part, lo, n_cols, sp if si else None, sp, rb, re - rb)
            be.sync()
            leaves = state.to_cpu().reshape(-1, 8)
            assert np.array_equal(leaves, tree.layer(log)), (mode, n_cols, log)
            t2 = be.merkle_from_leaves(state.ptr.value, log)
            assert np.array_equal(t2.root(), tree.root())
            for k in range(log):
                assert np.array_equal(t2.layer(k), tree.layer(k))
    finally:
        be.set_hash_mode(0)


# ---------------- error behaviour and edge cases (the reference surfaces Result<_, ProvingError>; allocation failures and
# bad arguments must come back as error codes with a message, never as silent garbage) --------------------------------

def test_argument_errors_are_reported(be, nz):
    tw = be.precompute_twiddles(6)
    cols = be.columns_from_host(rand_cols(1, 2, 9))
    with pytest.raises(nz.NexusHipError, match="twiddle"):          # domain larger than the twiddle tree
        be.interpolate_columns(tw, cols)
    with pytest.raises(nz.NexusHipError):
        be.precompute_twiddles(40)
    lde = be.columns_from_host(rand_cols(2, 20, 5))
    st = be.columns(8, 5)
    with pytest.raises(nz.NexusHipError, match="multiple of 16"):   # shard not aligned to a Blake2s block
        be.merkle_leaf_chain(lde, 8, 40, st.ptr.value, st.ptr.value)
    with pytest.raises(nz.NexusHipError, match="NULL exactly"):     # first shard must not take a state
        be.merkle_leaf_chain(lde, 0, 20, st.ptr.value, st.ptr.value)
    with pytest.raises(nz.NexusHipError):                            # synthetic machine needs >= 2 preprocessed columns
        be.prove([(6, 1, 4, 0)], nz.default_config())
    with pytest.raises(nz.NexusHipError):                            # blow-up 0 is not a valid PcsConfig
        be.prove([(6, 2, 4, 0)], nz.default_config(log_blowup=0))
    # the context is still usable after errors
    assert be.grind(np.zeros(8, np.uint32), 3) >= 0


def test_empty_and_single_column_inputs(be, oracle):
    tw = be.precompute_twiddles(13)
    # zero columns: every batched entry is a no-op
    be._chk(be.L.nx_interpolate_batch(be.ctx, tw.h, None, 0, 10))
    be._chk(be.L.nx_lde_batch(be.ctx, tw.h, None, 0, 10, 1, None))
    # one column goes through the pair kernel as a degenerate pair (and through the small kernel below 2^13)
    otw = oracle.Twiddles(14)
    for log in (5, 13):
        v = rand_cols(log + 300, 1, log)
        c = be.columns_from_host(v)
        out = be.lde(be.precompute_twiddles(log), c, 1)
        coeff = otw.interpolate(v[0])
        assert np.array_equal(c.to_cpu()[0], coeff)
        assert np.array_equal(out.to_cpu()[0], otw.evaluate(coeff, log + 1))
    # empty tree = Blake2s of nothing / zero state (reference: MerkleProver::commit(vec![]))
    for mode in (0, 1):
        be.set_hash_mode(mode)
        assert np.array_equal(be.merkle_commit([]).root(), oracle.merkle_commit([], mode))
    be.set_hash_mode(0)


def test_boundary_values_stay_canonical(be, oracle):
    """Columns made of 0, 1, p-1 and p-2 only: every sum/difference/product hits the reduction boundaries; the LDE must be
    canonical (< p, committed as raw words) and bit-exact."""
    log = 14
    rng = np.random.default_rng(99)
    vals = rng.choice(np.array([0, 1, P - 1, P - 2], np.uint32), size=(4, 1 << log))
    vals[0, :] = P - 1
    vals[1, :] = 0
    otw = oracle.Twiddles(log + 1)
    cols = be.columns_from_host(vals)
    lde = be.lde(be.precompute_twiddles(log), cols, 1)
    got_c, got_l = cols.to_cpu(), lde.to_cpu()
    assert got_c.max() < P and got_l.max() < P
    for c in range(4):
        coeff = otw.interpolate(vals[c])
        assert np.array_equal(got_c[c], coeff)
        assert np.array_equal(got_l[c], otw.evaluate(coeff, log + 1))


def test_config5_shaped_machine_is_accepted_by_the_oracle_verifier(be, nz, oracle):
    """BASELINE config #5 shape (keccak-precompile-like widths): two wide round components of different
    sizes with interaction trees ~2x wider than their main traces, plus small lookup tables — a mixed-degree commitment
    with four FRI column sizes.  (The oracle prover would take minutes at this width; its verifier is the check.)"""
    comps = [(14, 4, 260, 520), (13, 4, 200, 400), (12, 3, 9, 4), (11, 3, 7, 4)]
    cfg, ocfg = nz.default_config(pow_bits=8), O.default_cfg(pow_bits=8)
    w = be.prove(comps, cfg, seed=5150, ad=b"keccak-shaped")
    assert oracle.verify_synth(comps, ocfg, w, ad=b"keccak-shaped") is None
    w2 = w.copy(); w2[40] ^= 1
    assert oracle.verify_synth(comps, ocfg, w2, ad=b"keccak-shaped") is not None
    assert oracle.verify_synth(comps, ocfg, w, ad=b"other transcript") is not None


def test_constraint_violation_is_a_proving_error(be, nz):
    """ProvingError::ConstraintsNotSatisfied (reference core/src/lib.rs:22-24): committing a trace that violates the AIR
    must fail in the prover's own OODS check, not produce a proof.  The synthetic fill is always valid, so corrupt the
    check's other input instead: a PcsConfig whose constraint degree bound is too small for degree-2 constraints cannot
    be requested (log_constraint_degree >= 1 enforced), and a tampered proof is what the verifier tests cover; here we
    only pin the error code path of the argument check."""
    with pytest.raises(nz.NexusHipError):
        be.prove([(6, 2, 4, 0)], nz.default_config(log_constraint_degree=0))
    with pytest.raises(nz.NexusHipError, match="log_constraint_degree_bound exceeds"):
        be.prove([(6, 2, 4, 0, 2)], nz.default_config(log_constraint_degree=1))


@pytest.mark.parametrize("world,comps,kw", [
    (2, [(9, 20, 40, 24)], dict(pow_bits=6)),
    (4, [(10, 27, 100, 64)], dict(pow_bits=8)),
    (4, [(8, 3, 20, 0)], dict(pow_bits=5, log_constraint_degree=2)),       # constraint domain above the LDE: re-evaluation + a second all-to-all; no interaction tree
    (2, [(9, 18, 33, 17)], dict(pow_bits=6, hash_mode=1, fri_alpha_mode=1)),
    (8, [(8, 3, 5, 2)], dict(pow_bits=5)),                                    # fewer columns than ranks: some ranks transform nothing
    (2, [(14, 4, 20, 8), (11, 2, 9, 4)], dict(pow_bits=7)),                   # FRI layers large enough to stay row-sharded; two column sizes
    (4, [(13, 3, 18, 8), (13, 2, 7, 0), (9, 2, 3, 4)], dict(pow_bits=6, log_constraint_degree=2)),   # a run of two equal-size components shares one plan
    (8, [(12, 27, 347, 64)], dict(pow_bits=10)),
    (4, [(12, 4, 33, 8, 1), (9, 3, 17, 16, 2), (6, 2, 2, 0, 1)], dict(pow_bits=6, log_constraint_degree=2)),   # per-component bounds: only the small component is re-evaluated
])
def test_sharded_prove_is_bit_identical_to_single_gpu(nz, oracle, world, comps, kw):
    """SURVEY §8(e) / BASELINE config #4: ONE proof on `world` ranks (one context per rank on this GPU, threads, loopback transport):
    column-parallel LDE, one all-to-all per tree into row blocks, local leaf hashing / constraints / quotients / FRI folds, W subtree
    roots per tree.  Every rank must return the proof nx_prove_synth returns on one GPU — which the oracle proves too."""
    import threading
    from nexus_zkvm_amd.sharded import ThreadGroup
    cfg = nz.default_config(**kw)
    be0 = nz.HipBackend(0)
    ref = be0.prove(comps, cfg, seed=77, ad=b"shard")
    be0.close()
    assert oracle.verify_synth(comps, O.default_cfg(**kw), ref, ad=b"shard") is None
    group = ThreadGroup(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            be = nz.HipBackend(0)
            comm = nz.make_comm(rank, world, group.comm(rank, be))
            results[rank] = be.prove_sharded(comps, comm, cfg, seed=77, ad=b"shard")
            be.close()
        except Exception as e:   # noqa: BLE001
            errors.append((rank, repr(e)))
            try:
                group.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for r in range(world):
        assert results[r] is not None and np.array_equal(results[r], ref), (world, r)


# ---------------- "next" row R9: recorded AIR constraints on device (nx_eval_constraint_program) ----------------------------

def _random_program(ap, rng, n_cols, n_ops):
    """A random straight-line program that uses every opcode: base and secure arithmetic, masks at -1/0/+1, secure columns."""
    pb = ap.ProgramBuilder()
    base = [e for k in range(n_cols - 4) for e in pb.next_trace_mask(k, (0, 1) if k % 3 == 0 else (-1,) if k % 5 == 0 else (0,))]
    sec = pb.next_secure_mask(n_cols - 4, (0, 1))
    sec.append(pb.econst(rng.integers(0, P, 4)))
    for _ in range(n_ops):
        r = rng.integers(0, 8)
        if r < 3:
            a, b = base[rng.integers(len(base))], base[rng.integers(len(base))]
            base.append([a + b, a - b, a * b][r])
        elif r == 3:
            base.append(-base[rng.integers(len(base))] + int(rng.integers(0, P)))
        elif r < 6:
            a, b = sec[rng.integers(len(sec))], sec[rng.integers(len(sec))]
            sec.append([a + b, a * b][r - 4])
        elif r == 6:
            sec.append(sec[rng.integers(len(sec))] * base[rng.integers(len(base))] - sec[rng.integers(len(sec))])
        else:
            sec.append(base[rng.integers(len(base))] - sec[rng.integers(len(sec))] + base[rng.integers(len(base))])
        if rng.integers(0, 3) == 0:
            pb.add_constraint(base[-1] if rng.integers(0, 2) else sec[-1])
    pb.add_constraint(base[-1]); pb.add_constraint(sec[-1])
    return pb.build()


@pytest.mark.parametrize("log,lcd,seed", [(5, 1, 1), (6, 2, 2), (9, 1, 3)])
def test_constraint_program_matches_oracle(be, oracle, log, lcd, seed):
    import nexus_zkvm_amd.air_program as ap
    from test_air_program_cpu import denominators
    rng = np.random.default_rng(seed)
    e, n_cols = log + lcd, 12
    prog = _random_program(ap, rng, n_cols, 120)
    cols = rng.integers(0, P, (n_cols, 1 << e), dtype=np.uint32)
    pw = rng.integers(0, P, (prog.n_constraints, 4), dtype=np.uint32)
    den = denominators(log, e)
    start = rng.integers(0, P, (4, 1 << e), dtype=np.uint32)          # the accumulator is added to, not overwritten
    d_cols, acc = be.columns_from_host(cols), be.columns_from_host(start)
    ptrs = [d_cols.ptr.value + k * (4 << e) for k in range(n_cols)]
    be.eval_constraint_program(prog, ptrs, pw, den, log, e, acc)
    ref = oracle.eval_constraint_program(prog, list(cols), pw, den, log, e, acc4=list(start))
    assert np.array_equal(acc.to_cpu(), np.stack(ref))
    assert prog.n_regs <= 160


def test_constraint_program_reproduces_the_synthetic_machine(be, nz, oracle):
    """The synthetic AIR recorded through the generic evaluator gives, on the device's own LDE columns, the same accumulator as
    the oracle's interpreter, and a quotient that is a polynomial (the property that pins the oracle, now on GPU data)."""
    import nexus_zkvm_amd.air_program as ap
    from test_air_program_cpu import denominators, synthetic_program
    log, n_pre, n_main, n_inter = 10, 4, 37, 18
    comps = [(log, n_pre, n_main, n_inter)]
    e = log + 1
    tw = be.precompute_twiddles(e)
    ldes = []
    for tree in range(3):
        for s in be.synth_fill_tree(comps, tree, seed=5, inter_seed=99):
            ldes.append(be.lde(tw, s, 1))
    ptrs = [l.ptr.value + k * (4 << e) for l in ldes for k in range(l.n_cols)]
    host = np.concatenate([l.to_cpu() for l in ldes])
    prog = synthetic_program(ap, n_pre, n_main, n_inter)
    pw = np.random.default_rng(8).integers(0, P, (prog.n_constraints, 4), dtype=np.uint32)
    den = denominators(log, e)
    acc = be.columns(4, e)
    be._chk(be.L.nx_memset_zero(be.ctx, acc.ptr, C.c_size_t(4 << e)))
    be.eval_constraint_program(prog, ptrs, pw, den, log, e, acc)
    got = acc.to_cpu()
    assert np.array_equal(got, np.stack(oracle.eval_constraint_program(prog, list(host), pw, den, log, e)))
    be.interpolate_columns(tw, acc)
    coeffs = acc.to_cpu()
    assert not coeffs[:, (1 << log) + 1:].any() and coeffs[:, :1 << log].any()


def test_constraint_program_rejects_malformed_programs(be, nz):
    import nexus_zkvm_amd.air_program as ap
    pb = ap.ProgramBuilder()
    (a,) = pb.next_trace_mask(3)           # column 3 of a 2-column table
    pb.add_constraint(a)
    prog = pb.build()
    acc = be.columns(4, 5)
    cols = be.columns(2, 5)
    ptrs = [cols.ptr.value, cols.ptr.value + (4 << 5)]
    with pytest.raises(nz.NexusHipError, match="malformed"):
        be.eval_constraint_program(prog, ptrs, np.zeros((1, 4), np.uint32), np.ones(2, np.uint32), 4, 5, acc)
    with pytest.raises(nz.NexusHipError, match="alpha powers"):
        be.eval_constraint_program(prog, ptrs + ptrs, np.zeros((2, 4), np.uint32), np.ones(2, np.uint32), 4, 5, acc)


@pytest.mark.parametrize("log,lcd,seed", [(5, 1, 11), (6, 2, 12), (9, 1, 13), (12, 1, 14)])
def test_air_jit_matches_oracle_and_interpreter(be, oracle, log, lcd, seed):
    """nx_air_compile/nx_air_eval (hiprtc) == oracle == interpreter on random programs covering every opcode; one compiled kernel
    is reused with different lookup elements and alpha powers (they are run-time arguments)."""
    import nexus_zkvm_amd.air_program as ap
    from test_air_program_cpu import denominators
    rng = np.random.default_rng(seed)
    e, n_cols = log + lcd, 12
    prog = _random_program(ap, rng, n_cols, 150)
    kern = be.compile_air(prog, n_cols)
    den = denominators(log, e)
    for rep in range(2):
        cols = rng.integers(0, P, (n_cols, 1 << e), dtype=np.uint32)
        pw = rng.integers(0, P, (prog.n_constraints, 4), dtype=np.uint32)
        if rep:
            prog.econsts = rng.integers(0, P, np.asarray(prog.econsts).shape, dtype=np.uint32)
        start = rng.integers(0, P, (4, 1 << e), dtype=np.uint32)
        d_cols, acc, acc2 = be.columns_from_host(cols), be.columns_from_host(start), be.columns_from_host(start)
        ptrs = [d_cols.ptr.value + k * (4 << e) for k in range(n_cols)]
        kern.eval(ptrs, pw, den, log, e, acc)
        be.eval_constraint_program(prog, ptrs, pw, den, log, e, acc2)
        got = acc.to_cpu()
        assert np.array_equal(got, acc2.to_cpu())
        if log <= 9:
            assert np.array_equal(got, np.stack(oracle.eval_constraint_program(prog, list(cols), pw, den, log, e, acc4=list(start))))
    kern.close()


def test_air_jit_reproduces_the_synthetic_machine(be, nz, oracle):
    import nexus_zkvm_amd.air_program as ap
    from test_air_program_cpu import denominators, synthetic_program
    log, n_pre, n_main, n_inter = 10, 4, 37, 18
    e = log + 1
    tw = be.precompute_twiddles(e)
    ldes = []
    for tree in range(3):
        for s in be.synth_fill_tree([(log, n_pre, n_main, n_inter)], tree, seed=5, inter_seed=99):
            ldes.append(be.lde(tw, s, 1))
    ptrs = [l.ptr.value + k * (4 << e) for l in ldes for k in range(l.n_cols)]
    host = np.concatenate([l.to_cpu() for l in ldes])
    prog = synthetic_program(ap, n_pre, n_main, n_inter)
    pw = np.random.default_rng(8).integers(0, P, (prog.n_constraints, 4), dtype=np.uint32)
    den = denominators(log, e)
    acc = be.columns(4, e)
    be._chk(be.L.nx_memset_zero(be.ctx, acc.ptr, C.c_size_t(4 << e)))
    kern = be.compile_air(prog, len(ptrs))
    kern.eval(ptrs, pw, den, log, e, acc)
    assert np.array_equal(acc.to_cpu(), np.stack(oracle.eval_constraint_program(prog, list(host), pw, den, log, e)))
    with pytest.raises(nz.NexusHipError, match="log_size"):
        kern.eval(ptrs, pw, den[:1], log, log, acc)


def test_air_kernels_ahead_of_time_blob_and_cache_directory(be, nz, oracle, tmp_path):
    """VERDICT r4 weak #10 (the hiprtc compilation sat outside every number): (1) nx_air_kernel_save / nx_air_kernel_load — a compiled
    kernel as a blob, loaded by ANOTHER context without a compilation, same accumulator; a damaged blob is refused; (2) nx_air_cache_dir —
    the first prove of a machine in a fresh process compiles and stores, the first prove of the NEXT process loads every kernel from the
    directory (0 compilations) and returns the same proof bytes."""
    import json, subprocess, sys
    import nexus_zkvm_amd.air_program as ap
    from test_air_program_cpu import denominators
    rng = np.random.default_rng(77)
    log, e, n_cols = 8, 9, 12
    prog = _random_program(ap, rng, n_cols, 150)
    kern = be.compile_air(prog, n_cols)
    blob = kern.save()
    assert blob[:4] == b"NXAK" and len(blob) > 1000
    cols = rng.integers(0, P, (n_cols, 1 << e), dtype=np.uint32)
    pw = rng.integers(0, P, (prog.n_constraints, 4), dtype=np.uint32)
    den = denominators(log, e)
    start = rng.integers(0, P, (4, 1 << e), dtype=np.uint32)
    d_cols, acc = be.columns_from_host(cols), be.columns_from_host(start)
    kern.eval([d_cols.ptr.value + k * (4 << e) for k in range(n_cols)], pw, den, log, e, acc)
    want = acc.to_cpu()
    assert np.array_equal(want, np.stack(oracle.eval_constraint_program(prog, list(cols), pw, den, log, e, acc4=list(start))))
    b2 = nz.HipBackend(0)
    before = nz.air_cache_stats()
    k2 = nz.AirKernel(b2, prog, n_cols, blob=blob)
    assert nz.air_cache_stats()[0] == before[0]                        # no compilation
    d2, a2 = b2.columns_from_host(cols), b2.columns_from_host(start)
    k2.eval([d2.ptr.value + k * (4 << e) for k in range(n_cols)], pw, den, log, e, a2)
    assert np.array_equal(a2.to_cpu(), want)
    bad = bytearray(blob); bad[len(bad) // 2] ^= 1
    with pytest.raises(nz.NexusHipError, match="blob"):
        nz.AirKernel(b2, prog, n_cols, blob=bytes(bad))
    with pytest.raises(nz.NexusHipError, match="blob"):
        nz.AirKernel(b2, prog, n_cols, blob=blob[:100])
    k2.close(); b2.close(); kern.close()
    # the cache directory, across processes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json, time, numpy as np, torch, nexus_zkvm_amd as nz\n"
            "be = nz.HipBackend(0)\n"
            "comps = [(10, 4, 40, 24, 2, 1), (8, 3, 16, 32, 1, 5)]\n"
            "t0 = time.perf_counter(); w = be.prove_machine(comps, nz.default_config(pow_bits=4, log_constraint_degree=2), seed=9); dt = time.perf_counter() - t0\n"
            "print(json.dumps({'stats': nz.air_cache_stats(), 'first_prove_s': dt, 'sum': int(np.asarray(w, dtype=np.uint64).sum()), 'n': len(w)}))\n")
    env = dict(os.environ, NX_AIR_CACHE_DIR=str(tmp_path / "kernels"), PYTHONPATH=root)
    runs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
    (c1, h1, s1), (c2, h2, s2) = runs[0]["stats"], runs[1]["stats"]
    assert c1 >= 2 and h1 == 0 and s1 == c1, runs                       # compiled and stored
    assert c2 == 0 and h2 == c1 and s2 == 0, runs                       # the next process: every kernel from the directory
    assert runs[0]["sum"] == runs[1]["sum"] and runs[0]["n"] == runs[1]["n"]
    assert len(list((tmp_path / "kernels").glob("nxair-*.nxak"))) == c1


def test_many_kernel_program_compiled_in_helper_processes(be, nz, oracle, monkeypatch):
    """Round 6: a program of many kernels ("air.segment" small: >= 4 segments) is cut into one part per kernel, and the
    parts are compiled side by side by nx_air_cc helper PROCESSES (hiprtc serialises threads, not processes) — NX_AIR_COMPILE_PROCS=1
    compiles the same parts in this process.  The blob is the same bytes either way ("NXMM" container behind the header), a kernel loaded
    from it evaluates like the compiled one, and both match the oracle's interpreter."""
    import nexus_zkvm_amd.air_program as ap
    from test_air_program_cpu import denominators
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    helper = os.path.join(root, "nexus-zkvm_amd", "nx_air_cc")
    if not os.access(helper, os.X_OK):                     # built by csrc/Makefile next to libnexus_hip.so; a snapshot without it: build it here (host code only)
        import subprocess
        subprocess.run(["make", "-C", os.path.join(root, "nexus-zkvm_amd", "csrc"), "../nx_air_cc"], capture_output=True, timeout=120)
    if not os.access(helper, os.X_OK):
        pytest.skip("nx_air_cc is not next to the library and cannot be built here: every part compiles in-process (covered by the other AIR tests)")
    rng = np.random.default_rng(78)
    log, e, n_cols = 8, 9, 12
    prog = _random_program(ap, rng, n_cols, 400)
    cols = rng.integers(0, P, (n_cols, 1 << e), dtype=np.uint32)
    pw = rng.integers(0, P, (prog.n_constraints, 4), dtype=np.uint32)
    den = denominators(log, e)
    start = rng.integers(0, P, (4, 1 << e), dtype=np.uint32)
    want = np.stack(oracle.eval_constraint_program(prog, list(cols), pw, den, log, e, acc4=list(start)))
    blobs = []
    for procs in ("1", "8"):
        monkeypatch.setenv("NX_AIR_COMPILE_PROCS", procs)
        b = nz.HipBackend(0)
        b.set_option("air.segment", 300)
        kern = b.compile_air(prog, n_cols)
        blob = kern.save()
        assert blob[48:52] == b"NXMM" and int.from_bytes(blob[52:56], "little") >= 4        # one part per kernel
        d_cols, acc = b.columns_from_host(cols), b.columns_from_host(start)
        kern.eval([d_cols.ptr.value + k * (4 << e) for k in range(n_cols)], pw, den, log, e, acc)
        assert np.array_equal(acc.to_cpu(), want)
        k2 = nz.AirKernel(b, prog, n_cols, blob=blob)
        a2 = b.columns_from_host(start)
        k2.eval([d_cols.ptr.value + k * (4 << e) for k in range(n_cols)], pw, den, log, e, a2)
        assert np.array_equal(a2.to_cpu(), want)
        blobs.append(blob)
        k2.close(); kern.close(); b.close()
    assert blobs[0] == blobs[1]


def test_air_jit_rejects_malformed_programs(be, nz):
    import nexus_zkvm_amd.air_program as ap
    pb = ap.ProgramBuilder()
    (a,) = pb.next_trace_mask(3)
    pb.add_constraint(a)
    with pytest.raises(nz.NexusHipError, match="malformed"):
        be.compile_air(pb.build(), 2)


# ---------------- the prover session: stwo::prover::prove over recorded AIRs (nx_prover_*) ----------------------------------

def test_air_subset_kernels_add_up_to_the_whole_program(be, oracle):
    """nx_air_compile_subset: the kernel of the degree <= 3 constraints and the kernel of the others, run one after the other on the same
    accumulator, give the whole program's result bit for bit (same alpha powers, same column table; the columns a part does not
    read are passed as NULL), and nx_air_constraint_degrees names the split."""
    import nexus_zkvm_amd as nx
    from test_air_program_cpu import _mixed_degree_program, denominators
    prog, n_cols = _mixed_degree_program()
    deg = nx.air_constraint_degrees(prog, n_cols)
    low, high = (deg <= 3).astype(np.uint8), (deg > 3).astype(np.uint8)
    log, e = 7, 9
    rng = np.random.default_rng(21)
    cols = rng.integers(0, P, (n_cols, 1 << e), dtype=np.uint32)
    pw = rng.integers(0, P, (prog.n_constraints, 4), dtype=np.uint32)
    start = rng.integers(0, P, (4, 1 << e), dtype=np.uint32)
    den = denominators(log, e)
    d_cols = be.columns_from_host(cols)
    ptrs = [d_cols.ptr.value + k * (4 << e) for k in range(n_cols)]
    whole, parts = be.columns_from_host(start), be.columns_from_host(start)
    k_all, k_low, k_high = be.compile_air(prog, n_cols), be.compile_air(prog, n_cols, low), be.compile_air(prog, n_cols, high)
    k_all.eval(ptrs, pw, den, log, e, whole)
    p_low = list(ptrs); p_low[6] = None            # column 6 is read by degree-5 / degree-4 constraints only
    p_high = list(ptrs); p_high[3] = None          # column 3 by the cubic constraint only
    k_low.eval(p_low, pw, den, log, e, parts)
    k_high.eval(p_high, pw, den, log, e, parts)
    ref = np.stack(oracle.eval_constraint_program(prog, list(cols), pw, den, log, e, acc4=list(start)))
    assert np.array_equal(whole.to_cpu(), ref) and np.array_equal(parts.to_cpu(), ref)
    for k in (k_all, k_low, k_high):
        k.close()


@pytest.mark.parametrize("logs,bounds", [((10, 8), (2, 1)), ((6, 7), None)])
def test_degree_split_does_not_change_a_byte(nz, oracle, logs, bounds):
    """The degree-aware composition ("air.degree_split": constraints of degree <= 3 on the committed evaluations, the rest on the 4x
    domain, only the columns they read re-extended) gives the proof of the plain evaluation on the 4x domain — and of the oracle,
    which only knows the plain one — byte for byte, also when the +2 component has no high-degree constraint at all."""
    from test_prover_session_cpu import build_mixed_air
    ocfg = oracle.default_cfg(pow_bits=2, log_constraint_degree=2, log_blowup=1)
    cfg = _hip_cfg(nz, ocfg)
    for hd in (True, False):
        drive, _ = build_mixed_air(logs, lcd=2, bounds=bounds, high_degree=hd)
        so = oracle.ProverSession(ocfg, max(logs))
        ref = so.prove(drive(so, so.commit))
        for split in (1, 0):
            b = nz.HipBackend()
            b.set_option("air.degree_split", split)
            assert b.get_option("air.degree_split") == split
            sh = b.prover_session(cfg, max(logs))
            words = sh.prove(drive(sh, sh.commit))
            assert np.array_equal(words, ref), (hd, split)
            sh.close()
            b.close()


def _hip_cfg(nz, ocfg):
    return nz.default_config(pow_bits=int(ocfg[0]), log_blowup=int(ocfg[1]), n_queries=int(ocfg[2]), log_last_layer_degree_bound=int(ocfg[3]),
                             hash_mode=int(ocfg[4]), fri_alpha_mode=int(ocfg[5]), log_constraint_degree=int(ocfg[6]))


@pytest.mark.parametrize("comps,lcd", [([(6, 3, 20, 19)], 1), ([(5, 2, 18, 4), (7, 3, 5, 17)], 2), ([(10, 3, 40, 33)], 1), ([(6, 2, 3, 0)], 1)])
def test_session_proves_the_recorded_synthetic_machine_identically(be, nz, oracle, comps, lcd):
    """The synthetic machine as recorded components through nx_prover_* == nx_prove_synth == the oracle's built-in prove, byte
    for byte: composition via the JIT-compiled program, mask points from the recorded masks, OODS check via the host
    interpreter, and the session's own TreeBuilder/channel all agree with the hand-written path."""
    from test_prover_session_cpu import drive_synthetic, synthetic_components
    ocfg = oracle.default_cfg(pow_bits=3, log_constraint_degree=lcd, log_blowup=max(1, lcd))
    cfg = _hip_cfg(nz, ocfg)
    ad = b"\x07\x2a"
    ref = oracle.prove_synth(comps, ocfg, seed=9, ad=ad)
    s = be.prover_session(cfg, max(c[0] for c in comps))
    drive_synthetic(None, comps, 9, ad, s.commit, s)
    words = s.prove(synthetic_components(comps))
    assert np.array_equal(words, ref)
    assert np.array_equal(be.prove(comps, cfg, seed=9, ad=ad), ref)
    s.close()


@pytest.mark.parametrize("logs,lcd,bounds,hd", [((5, 7), 1, None, False), ((6,), 2, None, False), ((7, 5, 6), 1, None, False), ((11, 9), 1, None, False),
                                                ((12, 8, 10), 2, (1, 2, 1), False),    # per-component bounds: big +1 components, one small +2 component
                                                ((9, 9), 2, (2, 1), False),
                                                # degree-3 / degree-4 constraints under the +2 bound, base- and secure-field, interleaved with the
                                                # degree-2 ones: the prover evaluates the two parts on different domains (air.degree_split)
                                                ((10, 8), 2, (2, 1), True), ((7, 9, 6), 2, None, True)])
def test_session_logup_air_matches_the_oracle_session(be, nz, oracle, logs, lcd, bounds, hd):
    """A multi-component AIR with a secure logup column at offsets [-1, 0] and lookup elements drawn from the session's
    channel: same draws, same roots, same proof bytes as the CPU oracle's session, and the oracle's verifier accepts."""
    from test_prover_session_cpu import build_mixed_air
    ocfg = oracle.default_cfg(pow_bits=2, log_constraint_degree=lcd, log_blowup=lcd)
    cfg = _hip_cfg(nz, ocfg)
    drive, tree_logs = build_mixed_air(logs, lcd=lcd, bounds=bounds, high_degree=hd)
    so = oracle.ProverSession(ocfg, max(logs))
    oroots = []
    ocomps = drive(so, lambda cols: oroots.append(so.commit(cols)))
    ref = so.prove(ocomps)
    sh = be.prover_session(cfg, max(logs))
    hroots = []
    hcomps = drive(sh, lambda cols: hroots.append(sh.commit(cols)))
    assert all(np.array_equal(a, b) for a, b in zip(oroots, hroots))
    kernels = [be.compile_air(c.program, len(c.cols)) for c in hcomps]
    words, stats = sh.prove(hcomps, kernels=kernels, want_stats=True)
    assert np.array_equal(words, ref) and stats["total"] > 0
    assert np.array_equal(sh.digest(), so.digest())
    v = oracle.VerifierSession(ocfg)
    v.mix_u64(len(logs))
    v.commit(hroots[0], tree_logs[0]); v.commit(hroots[1], tree_logs[1])
    v.draw_felt(); v.draw_felt()
    for c in hcomps:
        v.mix_felts(np.asarray(c.program.econsts, np.uint32)[2])
    v.commit(hroots[2], tree_logs[2])
    assert v.verify(hcomps, words) is None
    sh.close()


@pytest.mark.parametrize("kw", [dict(hash_mode=1), dict(fri_alpha_mode=1), dict(log_blowup=2, n_queries=7), dict(log_last=2, pow_bits=0),
                                dict(log_blowup=3, log_constraint_degree=2, n_queries=2)])
def test_session_under_every_protocol_switch(be, nz, oracle, kw):
    """The session follows the same run-time switches as nx_prove_synth (Merkle node rule, FRI folding-alpha schedule, blow-up,
    query count, last-layer bound, constraint degree): byte-identical to the oracle session under each."""
    from test_prover_session_cpu import build_mixed_air
    logs = (6, 8)
    ocfg = oracle.default_cfg(**{"pow_bits": 2, **kw})
    cfg = _hip_cfg(nz, ocfg)
    drive, tree_logs = build_mixed_air(logs)
    so = oracle.ProverSession(ocfg, max(logs))
    ref = so.prove(drive(so, so.commit))
    sh = be.prover_session(cfg, max(logs))
    words = sh.prove(drive(sh, sh.commit))
    assert np.array_equal(words, ref)
    sh.close()


def test_session_refuses_invalid_traces_and_misuse(be, nz, oracle):
    from test_prover_session_cpu import build_mixed_air
    import nexus_zkvm_amd.air_program as ap
    cfg = _hip_cfg(nz, oracle.default_cfg(pow_bits=2))
    drive, _ = build_mixed_air((5,))
    s = be.prover_session(cfg, 5)
    comps = drive(s, s.commit, tamper="main")
    with pytest.raises(nz.NexusHipError, match="ConstraintsNotSatisfied"):
        s.prove(comps)
    s = be.prover_session(cfg, 5)
    comps = drive(s, s.commit)
    c0 = comps[0]
    with pytest.raises(nz.NexusHipError, match="claimed by no component"):
        s.prove([ap.Component(c0.log_size, c0.program, c0.cols[:-1], c0.masks[:-1])])
    with pytest.raises(nz.NexusHipError, match="missing from the column's mask"):
        s.prove([ap.Component(c0.log_size, c0.program, c0.cols, [[0]] * len(c0.cols))])
    with pytest.raises(nz.NexusHipError, match="log_constraint_degree_bound exceeds"):   # the twiddle tree is sized by the config's bound (1 here)
        s.prove([ap.Component(c0.log_size, c0.program, c0.cols, c0.masks, log_constraint_degree_bound=2)])
    first = s.prove(comps)                               # the session is still usable after the refused calls ...
    assert len(first) > 0 and np.array_equal(s.prove(comps), first)   # ... and after a proof: the composition tree and the channel are put back
    with pytest.raises(nz.NexusHipError, match="trace trees are fixed"):
        s.tree_begin([5])                               # a session that has proved does not take further trees
    s = be.prover_session(cfg, 5)
    s.commit([np.zeros(32, np.uint32)])
    with pytest.raises(nz.NexusHipError, match="outside the committed trees"):
        s.prove(comps)
    s.tree_begin([5, 5])
    with pytest.raises(nz.NexusHipError, match="not committed"):
        s.tree_begin([5])
    with pytest.raises(nz.NexusHipError, match="log size"):
        be.prover_session(cfg, 5).tree_begin([6])


@pytest.mark.parametrize("log,k", [(4, 3), (12, 5), (13, 2), (17, 3)])
def test_logup_finalize_last_batch_matches_oracle(be, oracle, log, k):
    """Several secure columns finalised in one call (block scans, device scan of the block totals, fix-up) == the oracle's
    finalize_last on each; sizes below, at and above one 4096-row scan block and across many blocks."""
    rng = np.random.default_rng(log * 10 + k)
    cols = [rng.integers(0, P, (4, 1 << log), dtype=np.uint32) for _ in range(k)]
    dev = [be.columns_from_host(c) for c in cols]
    claimed = be.logup_finalize_last_batch(dev)
    for c, d, cs in zip(cols, dev, claimed):
        ref_cols, ref_sum = oracle.logup_finalize_last([x.copy() for x in c])
        assert np.array_equal(cs, ref_sum) and np.array_equal(d.to_cpu(), np.stack(ref_cols))


@pytest.mark.parametrize("log", [6, 12])
def test_session_end_to_end_with_device_logup(be, nz, oracle, log):
    """Ranks 1 + 2 + the session together, nothing but the main trace and the lookup elements on the host: the interaction column
    is generated by nx_logup_col / nx_logup_finalize_last straight into the session's interaction tree, the claimed sum it returns
    feeds the recorded constraint's shift, and the proof verifies in the oracle's independent verifier session."""
    import nexus_zkvm_amd.air_program as ap
    import air_examples as X
    ocfg = oracle.default_cfg(pow_bits=2)
    cfg = _hip_cfg(nz, ocfg)
    n = 1 << log
    nat, fin = X.logup_main_trace(log, seed=21)
    s = be.prover_session(cfg, log)
    s.mix_u64(log)
    r0 = s.commit([np.zeros(n, np.uint32)])
    mptrs = s.tree_begin([log] * 3)
    assert mptrs[1] - mptrs[0] == 4 * n                                   # one slab: the three main columns are contiguous
    main = nz.DeviceColumns.view(be, mptrs[0], 3, log).upload(np.stack(fin))
    tuple_cols = be.clone_columns(nz.DeviceColumns.view(be, mptrs[0], 2, log))      # (a, b): the commit turns the tree's copy into coefficients
    r1 = s.tree_commit()
    z, alpha = s.draw_felt(), s.draw_felt()
    iptrs = s.tree_begin([log] * 4)
    S = nz.DeviceColumns.view(be, iptrs[0], 4, log)
    # frac = 1 / (z - a - alpha b) = (-1) / (a + alpha b - z): Relation::combine gives the bracket, scale = -1
    be.logup_col({"tuple": tuple_cols, "alphas": np.array([[1, 0, 0, 0], alpha], np.uint32), "z": z, "scale": (P - 1, 0, 0, 0)}, out=S)
    claimed = be.logup_finalize_last(S)
    shift = X.qmul(claimed, [pow(n, P - 2, P), 0, 0, 0])
    ref_cols, ref_shift = X.logup_interaction_trace(log, nat, z, alpha) if log <= 8 else (None, None)
    if ref_cols is not None:
        assert np.array_equal(S.to_cpu(), np.stack(ref_cols)) and np.array_equal(shift, ref_shift)
    s.mix_felts(claimed)
    r2 = s.tree_commit()
    comp = X.logup_component(ap, log, z, alpha, shift)
    comp = ap.Component(log, comp.program, comp.cols + [(0, 0)], comp.masks + [[0]])
    words = s.prove([comp])
    v = oracle.VerifierSession(ocfg)
    v.mix_u64(log)
    v.commit(r0, [log]); v.commit(r1, [log] * 3)
    assert np.array_equal(v.draw_felt(), z) and np.array_equal(v.draw_felt(), alpha)
    v.mix_felts(claimed)
    v.commit(r2, [log] * 4)
    assert v.verify([comp], words) is None
    s.close()


@pytest.mark.parametrize("n_trees", [2, 4])
def test_session_with_two_and_four_trace_trees(be, nz, oracle, n_trees):
    """The device session proves the tree count it is given (the reference: three; VERDICT r3 hygiene: no three-tree literal): the
    proofs of a 2-tree and a 4-tree statement equal the oracle session's word for word."""
    import nexus_zkvm_amd.air_program as ap
    import air_examples as X
    kw = dict(pow_bits=3)
    cfg, ocfg = nz.default_config(**kw), oracle.default_cfg(**kw)
    trees, comp = X.tree_count_statement(ap, n_trees)
    so = oracle.ProverSession(ocfg, comp.log_size)
    so.mix_u64(n_trees)
    oroots = [so.commit(t) for t in trees]
    ref = so.prove([comp])
    sh = be.prover_session(cfg, comp.log_size)
    sh.mix_u64(n_trees)
    hroots = [sh.commit(t) for t in trees]
    assert all(np.array_equal(a, b) for a, b in zip(oroots, hroots))
    assert np.array_equal(sh.prove([comp]), ref)
    sh.close()


def test_c_example_proves_through_the_session(tmp_path):
    """examples/session_prove.c: the session driven from plain C99 — proves, and refuses a trace that violates the constraint."""
    import shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "nexus-zkvm_amd")
    exe = str(tmp_path / "session_prove")
    subprocess.run([shutil.which("gcc"), "-std=c99", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "session_prove.c"),
                    "-L" + lib_dir, "-lnexus_hip", "-Wl,-rpath," + lib_dir, "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok ") and int(out.split()[1]) > 100
    out = subprocess.run([exe, "bad"], check=True, capture_output=True, text=True).stdout
    assert out.startswith("refused:") and "ConstraintsNotSatisfied" in out


# ---------------- "next" row R8: logup interaction trace on device --------------------------------------------------------

@pytest.mark.parametrize("log", [4, 9, 13, 16])
def test_logup_pipeline_matches_oracle(be, oracle, log):
    """A small interaction trace in the reference's shape: column 0 = one fraction 1/combine([limb]) (v1, range256.rs:271-288),
    column 1 = two merged fractions with multiplicity numerators on top of column 0 (v2, logup_trace_builder.rs:86-101),
    then finalize_last.  Every intermediate column and the claimed sum bit-exact against the oracle."""
    rng = np.random.default_rng(100 + log)
    n = 1 << log
    tuple_cols = rng.integers(0, P, (5, n), dtype=np.uint32)
    alphas = rng.integers(0, P, (5, 4), dtype=np.uint32)
    z = rng.integers(0, P, 4, dtype=np.uint32)
    mult = rng.integers(0, 1 << 16, (2, n), dtype=np.uint32)
    d_tuple = be.columns_from_host(tuple_cols)
    d_mult0, d_mult1 = be.columns_from_host(mult[0]), be.columns_from_host(mult[1])

    den_a = be.logup_combine(be.columns_from_host(tuple_cols[:1]), alphas[:1], z)
    ref_den_a = oracle.logup_combine(list(tuple_cols[:1]), alphas[:1], z)
    assert np.array_equal(den_a.to_cpu(), np.stack(ref_den_a))
    den_b = be.logup_combine(d_tuple, alphas, z)
    ref_den_b = oracle.logup_combine(list(tuple_cols), alphas, z)
    assert np.array_equal(den_b.to_cpu(), np.stack(ref_den_b))

    col0 = be.logup_finalize_col(den_a)                                           # 1 / denom
    ref0 = oracle.logup_finalize_col(ref_den_a)
    assert np.array_equal(col0.to_cpu(), np.stack(ref0))
    minus_one = (P - 1, 0, 0, 0)
    col1 = be.logup_finalize_col(den_a, scale_a=minus_one, mult_a=d_mult0, den_b=den_b, scale_b=(1, 0, 0, 0), mult_b=d_mult1, prev=col0)
    ref1 = oracle.logup_finalize_col(ref_den_a, scale_a=minus_one, mult_a=mult[0], den_b=ref_den_b, mult_b=mult[1], prev=ref0)
    assert np.array_equal(col1.to_cpu(), np.stack(ref1))

    # fused form: same columns without materialising the denominators
    fa = dict(tuple=be.columns_from_host(tuple_cols[:1]), alphas=alphas[:1], z=z)
    f0 = be.logup_col(fa)
    assert np.array_equal(f0.to_cpu(), np.stack(ref0))
    f1 = be.logup_col(dict(fa, mult=d_mult0, scale=minus_one), dict(tuple=d_tuple, alphas=alphas, z=z, mult=d_mult1), prev=f0)
    assert np.array_equal(f1.to_cpu(), np.stack(ref1))

    # batched form: every column of the component in one launch (a third fraction with a wide tuple on top)
    wide = dict(tuple=d_tuple, alphas=alphas, z=z, mult=d_mult1, scale=minus_one)
    b0, b1, b2 = be.logup_cols([fa, dict(fa, mult=d_mult0, scale=minus_one), wide])
    assert np.array_equal(b0.to_cpu(), np.stack(ref0))
    ref_b1 = oracle.logup_finalize_col(ref_den_a, scale_a=minus_one, mult_a=mult[0], prev=ref0)
    assert np.array_equal(b1.to_cpu(), np.stack(ref_b1))
    assert np.array_equal(b2.to_cpu(), np.stack(oracle.logup_finalize_col(ref_den_b, scale_a=minus_one, mult_a=mult[1], prev=ref_b1)))

    claimed = be.logup_finalize_last(col1)
    ref_last, ref_claimed = oracle.logup_finalize_last(ref1)
    assert np.array_equal(claimed, ref_claimed)
    got = col1.to_cpu()
    assert np.array_equal(got, np.stack(ref_last)) and got.max() < P


@pytest.mark.parametrize("log", [5, 12])
def test_logup_cols_batched_matches_oracle(be, oracle, log):
    """nx_logup_cols_batched = finalize_logup_batched on the trace side: relation tuples of 1 ... 8 columns with their alpha powers
    (Relation::combine), fractions assigned to batches in pairs (NULL batching, odd count), by an explicit monotone batching with a
    batch of three, and by a NON-monotone one (Stwo's batching is any assignment): column j = the sum of the fractions of batches
    <= j.  The oracle merges a pair as LogupTraceBuilder does, (a d + b c) / (b d) (reference prover2/machine/src/lookups/
    logup_trace_builder.rs:93-97), and chains single fractions through finalize_col's `prev`."""
    rng = np.random.default_rng(700 + log)
    n = 1 << log
    widths = [1, 2, 3, 4, 5, 6, 7, 8, 2, 1, 3]
    F = len(widths)
    z = rng.integers(0, P, 4, dtype=np.uint32)
    alpha = rng.integers(0, P, 4, dtype=np.uint32)
    ap = [np.array([1, 0, 0, 0], np.uint32)]
    for _ in range(7):
        ap.append(oracle.qm31_mul(ap[-1], alpha))
    ap = np.stack(ap)
    tup = [rng.integers(0, P, (w, n), dtype=np.uint32) for w in widths]
    mult = [rng.integers(0, 1 << 20, n, dtype=np.uint32) if f % 3 != 1 else None for f in range(F)]
    scale = [(P - 1, 0, 0, 0) if f % 2 else (1, 0, 0, 0) for f in range(F)]
    fracs, keep = [], []
    for f in range(F):
        d = dict(tuple=be.columns_from_host(tup[f]), alphas=ap[:widths[f]], z=z, scale=scale[f])
        if mult[f] is not None:
            d["mult"] = be.columns_from_host(mult[f])
        fracs.append(d)
    dens = [oracle.logup_combine(list(tup[f]), ap[:widths[f]], z) for f in range(F)]

    def expected(batching):
        n_cols = max(batching) + 1
        cols, prev = [], None
        for j in range(n_cols):
            fs = [f for f in range(F) if batching[f] == j]
            while fs:
                if len(fs) >= 2:
                    a, b = fs[0], fs[1]; fs = fs[2:]
                    prev = oracle.logup_finalize_col(dens[a], scale_a=scale[a], mult_a=mult[a], den_b=dens[b], scale_b=scale[b], mult_b=mult[b], prev=prev)
                else:
                    a = fs[0]; fs = fs[1:]
                    prev = oracle.logup_finalize_col(dens[a], scale_a=scale[a], mult_a=mult[a], prev=prev)
            cols.append(prev)
        return cols

    for batching in (None, [0, 0, 1, 2, 2, 2, 3, 4, 4, 5, 5], [3, 0, 1, 1, 0, 2, 4, 4, 2, 3, 0]):
        got = be.logup_cols_batched(fracs, batching)
        want = expected(batching if batching is not None else [f // 2 for f in range(F)])
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert np.array_equal(g.to_cpu(), np.stack(w))
    # every batch 0 .. last must hold a fraction (finalize_logup_batched asserts it); a batch index outside the columns is refused
    import nexus_zkvm_amd as nz
    with pytest.raises(nz.NexusHipError):
        be.logup_cols_batched(fracs, [0, 0, 2, 2, 2, 2, 3, 4, 4, 5, 5])
    with pytest.raises(nz.NexusHipError):
        be.logup_cols_batched(fracs, [0] * F, n_cols=0)


@pytest.mark.parametrize("log", [6, 13])
def test_logup_cols_staged_reads_equal_the_direct_ones_and_the_oracle(be, nz, oracle, log):
    """nx_logup_cols requests every read of a group of 8 fractions up front and parks the values in LDS ("logup.staged", default on) when
    the group's tuples are at most 32 columns wide, and reads where it uses them otherwise.  Groups of 21 columns (three request rounds),
    of exactly 32, of 33 (the direct path) and a ragged last group of 3 fractions, with and without multiplicities, secure and base
    numerators: staged == direct ("logup.staged" = 0) == the oracle's LogupColGenerator chain, every column."""
    rng = np.random.default_rng(4100 + log)
    n = 1 << log
    widths = [3, 1, 2, 4, 1, 2, 3, 5,   4, 4, 4, 4, 4, 4, 4, 4,   5, 4, 4, 4, 4, 4, 4, 4,   8, 1, 2]
    F = len(widths)
    z, alpha = rng.integers(0, P, 4, dtype=np.uint32), rng.integers(0, P, 4, dtype=np.uint32)
    ap = [np.array([1, 0, 0, 0], np.uint32)]
    for _ in range(7):
        ap.append(oracle.qm31_mul(ap[-1], alpha))
    ap = np.stack(ap)
    tup = [rng.integers(0, P, (w, n), dtype=np.uint32) for w in widths]
    mult = [rng.integers(0, 1 << 20, n, dtype=np.uint32) if f % 3 != 1 else None for f in range(F)]
    scale = [(P - 1, 0, 0, 0) if f % 2 else ((5, 6, 7, 8) if f % 5 == 0 else (1, 0, 0, 0)) for f in range(F)]
    want, prev = [], None
    for f in range(F):
        prev = oracle.logup_finalize_col(oracle.logup_combine(list(tup[f]), ap[:widths[f]], z), scale_a=scale[f], mult_a=mult[f], prev=prev)
        want.append(np.stack(prev))
    direct = nz.HipBackend(0)
    direct.set_option("logup.staged", 0)
    for b in (be, direct):
        fracs = []
        for f in range(F):
            d = dict(tuple=b.columns_from_host(tup[f]), alphas=ap[:widths[f]], z=z, scale=scale[f])
            if mult[f] is not None:
                d["mult"] = b.columns_from_host(mult[f])
            fracs.append(d)
        got = b.logup_cols(fracs)
        for f in range(F):
            assert np.array_equal(got[f].to_cpu(), want[f]), (b is be, f)
        pairs = b.logup_cols_batched(fracs)                        # pairs: the running sum after every second fraction (and after the odd last one)
        for j, g in enumerate(pairs):
            assert np.array_equal(g.to_cpu(), want[min(2 * j + 1, F - 1)]), (b is be, j)
    direct.close()


@pytest.mark.parametrize("log,batching,segment", [(6, "pairs", 9000), (12, "single", 9000), (12, [2, 0, 1, 1, 0], 200), (15, "pairs", 300)])
def test_logup_program_matches_oracle(be, nz, oracle, log, batching, segment):
    """nx_logup_program: the interaction trace of a component from the relation entries its recorded AIR declares —
    tuples of 1 ... 3 values that are columns, expressions (a + 5) or next-row reads, multiplicities 1, -m and (q - 1), two relations —
    compiled by hiprtc, against the oracle's literal interpreter; with a small "air.segment" budget the program is several kernels,
    each continuing from the running sum its predecessor stored; then finalize_last."""
    import nexus_zkvm_amd.air_program as ap
    import air_examples as AE
    rng = np.random.default_rng(log)
    z, alpha = rng.integers(0, P, 4, dtype=np.uint32), rng.integers(0, P, 4, dtype=np.uint32)
    nat, fin = AE.relation_main_trace(log, 500 + log)
    frac = AE.relation_program(ap, z, alpha, (0, 0, 0, 0), batching).build_logup()
    want = oracle.logup_program(frac, fin + [None] * (4 * frac.n_logup_cols), log, frac.n_logup_cols)
    b = nz.HipBackend(0)
    b.set_option("air.segment", segment)
    d = b.columns_from_host(np.stack(fin))
    ptrs = [d.ptr.value + k * (4 << log) for k in range(len(fin))] + [None] * (4 * frac.n_logup_cols)
    got = b.logup_program(frac, ptrs, log)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert np.array_equal(g.to_cpu(), np.stack(w))
    claimed = b.logup_finalize_last(got[-1])
    last, ref_claimed = oracle.logup_finalize_last(want[-1])
    assert np.array_equal(claimed, ref_claimed) and np.array_equal(got[-1].to_cpu(), np.stack(last))
    # a second call reuses the compiled kernels (other lookup elements: they are run-time constants)
    frac2 = AE.relation_program(ap, alpha, z, (0, 0, 0, 0), batching).build_logup()
    assert np.array_equal(np.asarray(frac2.instrs), np.asarray(frac.instrs))
    c0 = nz.air_cache_stats()[0]
    got2 = b.logup_program(frac2, ptrs, log)
    assert nz.air_cache_stats()[0] == c0
    want2 = oracle.logup_program(frac2, fin + [None] * (4 * frac.n_logup_cols), log, frac.n_logup_cols)
    assert all(np.array_equal(g.to_cpu(), np.stack(w)) for g, w in zip(got2, want2))
    with pytest.raises(nz.NexusHipError, match="NULL"):
        b.logup_program(frac, [None] * len(ptrs), log)
    b.close()


def test_known_answers_of_the_reference_dump_on_the_device(be, nz, oracle):
    """tools/replay_reference_dump.py's two logup known answers ("logup_pairs": LogupTraceGenerator in pairs; "logup_wide": the 200-element
    state relation with a constant entry, a column-sum entry and the numerator (m - 1)) as the DEVICE computes them — nx_logup_cols_batched
    resp. nx_logup_program — against the oracle's literal evaluation.  On a box with cargo the same functions are compared with what
    tools/dump_reference.rs prints from Stwo's own LookupElements::combine / LogupTraceGenerator."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("replay_reference_dump", os.path.join(root, "tools", "replay_reference_dump.py"))
    rp = importlib.util.module_from_spec(spec); spec.loader.exec_module(rp)
    rng = np.random.default_rng(200)
    z, alpha = rng.integers(1, P, 4, dtype=np.uint32), rng.integers(1, P, 4, dtype=np.uint32)
    for fn in (rp.logup_pairs_columns, rp.logup_wide_columns):
        want, wclaimed = fn(z, alpha)
        got, gclaimed = fn(z, alpha, be)
        assert len(want) == len(got) and all(np.array_equal(a, b) for a, b in zip(want, got)) and np.array_equal(wclaimed, gclaimed), fn.__name__


def test_session_with_the_interaction_trace_generated_from_the_recorded_air(be, nz, oracle):
    """The whole flow a Rust-side prove takes once the chips' generators are gone (reference_patch/machine_hip.rs): the main tree
    goes up from host memory with its evaluations KEPT on the device, the lookup elements are drawn, nx_logup_program + finalize_last
    build the interaction tree on the device from the recorded relation entries, the claimed sum is mixed, the tree committed from
    DEVICE columns, and the proof equals the oracle session's, word for word."""
    import nexus_zkvm_amd.air_program as ap
    import air_examples as AE
    from test_logup_cpu import _relation_statement
    log = 10
    for batching in ("pairs", "single"):
        ref, frac, (z0, a0), fin = _relation_statement(log, batching, seed=77)
        cfg = nz.default_config(pow_bits=3)
        s = be.prover_session(cfg, log)
        s.mix_u64(log)
        s.commit([])
        _, kept = s.commit_host(fin, keep=range(len(fin)))
        z, alpha = s.draw_felts(2)
        assert np.array_equal(z, z0) and np.array_equal(alpha, a0)
        prog = AE.relation_program(ap, z, alpha, (0, 0, 0, 0), batching).build_logup()
        ptrs = [kept[k].ptr.value for k in range(len(fin))] + [None] * (4 * prog.n_logup_cols)
        cols = be.logup_program(prog, ptrs, log)
        claimed = be.logup_finalize_last(cols[-1])
        n_inv = pow((1 << log) % P, P - 2, P)
        shift = [(int(x) * n_inv) % P for x in claimed]
        s.mix_felts(np.array([claimed], np.uint32))
        dev = s.tree_begin([log] * (4 * prog.n_logup_cols))
        for j, col in enumerate(cols):
            for q in range(4):
                be._chk(be.L.nx_copy(be.ctx, C.c_void_p(dev[4 * j + q]), C.c_void_p(col.ptr.value + q * (4 << log)), C.c_size_t(1 << log)))
        s.tree_commit()
        pb = AE.relation_program(ap, z, alpha, shift, batching)
        words = s.prove([AE.relation_component(ap, log, pb, prog.n_logup_cols)])
        s.close()
        assert np.array_equal(words, ref)


def test_prover2_shaped_session_builds_every_interaction_trace_in_place(be, nz, oracle):
    """The Python twin of rust/nexus-hip/reference_patch/prove2_hip.rs (reference prover2/machine/src/prove.rs:34-135) and of
    simd_host.rs interaction_tree_on_device: components of their own log sizes; the preprocessed and the main tree go up from HOST
    memory, mixed sizes in one tree, every column's evaluations kept; the lookup elements are drawn; every component's fraction program
    writes its logup columns STRAIGHT into the session's interaction-tree columns (tree_begin) — pairs and single-fraction columns side by
    side, nothing copied —, finalize_last in place, the claimed sums mixed, the tree committed; the proof is the oracle session's."""
    import nexus_zkvm_amd.air_program as ap
    import air_examples as AE
    from test_logup_cpu import _v2_statement, _v2_trees, _v2_component_cols, V2_MAIN_POS
    logs, batchings = (9, 11, 10), ("pairs", "single", "pairs")
    ref, _, oroots, tree_logs, _, oclaimed = _v2_statement(logs, batchings)
    cfg = nz.default_config(pow_bits=3)
    fins, tree0, tree1 = _v2_trees(logs, 31)
    s = be.prover_session(cfg, max(logs))
    for log in logs:
        s.mix_u64(log)
    r0, kept0 = s.commit_host(tree0, keep=range(len(tree0)))
    r1, kept1 = s.commit_host(tree1, keep=range(len(tree1)))
    assert np.array_equal(r0, oroots[0]) and np.array_equal(r1, oroots[1])
    z, alpha = s.draw_felts(2)
    fracs = [AE.relation_program(ap, z, alpha, (0, 0, 0, 0), b).build_logup() for b in batchings]
    dev = s.tree_begin(tree_logs[2])
    claimed, shifts, base = [], [], 0
    for c, (log, frac) in enumerate(zip(logs, fracs)):
        ptrs = [kept0[c].ptr.value if k == 1 else kept1[6 * c + V2_MAIN_POS[k]].ptr.value for k in range(7)] + [None] * (4 * frac.n_logup_cols)
        out = dev[base:base + 4 * frac.n_logup_cols]
        assert be.logup_program(frac, ptrs, log, out_ptrs=out) is out
        claimed.append(be.logup_finalize_last(out[-4:], log_size=log))
        n_inv = pow((1 << log) % P, P - 2, P)
        shifts.append([(int(x) * n_inv) % P for x in claimed[-1]])
        base += 4 * frac.n_logup_cols
    assert np.array_equal(np.array(claimed, np.uint32), oclaimed)
    s.mix_felts(np.array(claimed, np.uint32))
    assert np.array_equal(s.tree_commit(), oroots[2])
    comps, base = [], 0
    for c, (log, b, frac) in enumerate(zip(logs, batchings, fracs)):
        comps.append(ap.Component(log, AE.relation_program(ap, z, alpha, shifts[c], b).build(), _v2_component_cols(c, frac.n_logup_cols, base)))
        base += 4 * frac.n_logup_cols
    words = s.prove(comps)
    s.close()
    assert np.array_equal(words, ref)
    with pytest.raises(nz.NexusHipError, match="output columns"):
        be.logup_program(fracs[0], [None] * 19, logs[0], out_ptrs=[1, 2, 3])


def test_preprocessed_tree_shared_between_sessions_and_proofs(be, nz, oracle):
    """the preprocessed tree of a program is the same in every proof of that program (reference machine.rs:208-228)
    and in every verification (machine.rs:363-417).  nx_prover_tree_share / nx_prover_tree_adopt: a second session adopts the tree the
    first one committed — no upload, transform or hashing — and its proof equals the oracle's for the same statement, word for word;
    the first session still proves; a tree of another blowup is refused.  And nx_prove_machine's "machine.reuse_preprocessed" keeps the
    tree per statement shape: later proofs (other seeds) equal the ones a fresh context produces."""
    import nexus_zkvm_amd.air_program as ap
    import air_examples as AE
    log, n_pre, n_main, n_inter = 9, 3, 12, 6
    kw = dict(pow_bits=4)
    cfg, ocfg = nz.default_config(**kw), O.default_cfg(**kw)
    comps = [(log, n_pre, n_main, n_inter)]
    comp = AE.synthetic_component(ap, log, n_pre, n_main, n_inter)
    pre = [c for s in be.synth_fill_tree(comps, 0, 1) for c in s.to_cpu()]

    def statement(session, seed, first_tree):
        session.mix_u64(log)
        first_tree(session)
        main = oracle.synth_tree_columns(comps, 1, seed)
        session.commit(main)
        session.mix_felts(np.zeros(4, np.uint32))
        session.commit(oracle.synth_tree_columns(comps, 2, seed, inter_seed=5))
        return session.prove([comp])

    refs = {}
    for seed in (11, 12):
        o = O.ProverSession(ocfg, log)
        refs[seed] = statement(o, seed, lambda s: s.commit(pre))
    s1 = be.prover_session(cfg, log)
    w1 = statement(s1, 11, lambda s: s.commit(pre))
    assert np.array_equal(w1, refs[11])
    shared = s1.share_tree(0)
    root, n = shared.root()
    assert n == n_pre and np.array_equal(root, w1[6:14])
    s2 = be.prover_session(cfg, log)
    w2 = statement(s2, 12, lambda s: s.adopt_tree(shared))
    assert np.array_equal(w2, refs[12])
    assert np.array_equal(s1.prove([comp]), refs[11])                       # the sharing session still proves (its entry is a view now)
    s1.close()                                                            # ... and may go: the handle and s2 keep the tree alive
    s3 = be.prover_session(cfg, log)
    assert np.array_equal(statement(s3, 11, lambda s: s.adopt_tree(shared)), refs[11])
    s4 = be.prover_session(nz.default_config(pow_bits=4, log_blowup=2), log)
    s4.mix_u64(log)
    with pytest.raises(nz.NexusHipError, match="blowup"):
        s4.adopt_tree(shared)
    other = nz.HipBackend(0)
    s5 = other.prover_session(cfg, log)
    with pytest.raises(nz.NexusHipError, match="another context"):
        s5.adopt_tree(shared)
    s5.close(); other.close()
    shared.release(); s2.close(); s3.close(); s4.close()
    # nx_prove_machine
    mcomps = [(11, 27, 60, 16), (8, 3, 9, 4, 1, 1)]
    mcfg = nz.default_config(pow_bits=5)
    fresh = [be.prove_machine(mcomps, mcfg, seed=s, ad=b"r") for s in (1, 2, 3)]
    b = nz.HipBackend(0)
    b.set_option("machine.reuse_preprocessed", 1)
    for s, want in zip((1, 2, 3, 1), fresh + fresh[:1]):
        assert np.array_equal(b.prove_machine(mcomps, mcfg, seed=s, ad=b"r"), want)
    assert np.array_equal(b.prove_machine([(10, 27, 60, 16)], mcfg, seed=1), be.prove_machine([(10, 27, 60, 16)], mcfg, seed=1))   # another shape: its own tree
    b.close()


def test_column_utilities(be):
    """nx_copy (Column::clone) and the building blocks of a modular all-reduce (nx_m31_add_into / widen / narrow)."""
    rng = np.random.default_rng(3)
    a = rng.integers(0, P, (3, 1 << 10), dtype=np.uint32)
    b = rng.integers(0, P, (3, 1 << 10), dtype=np.uint32)
    a[0, :4] = [0, P - 1, 1, P - 1]; b[0, :4] = [0, P - 1, P - 1, 1]
    da, db = be.columns_from_host(a), be.columns_from_host(b)
    dc = be.clone_columns(da)
    assert np.array_equal(dc.to_cpu(), a)
    n = a.size
    be._chk(be.L.nx_m31_add_into(be.ctx, dc.ptr, db.ptr, C.c_size_t(n)))
    assert np.array_equal(dc.to_cpu(), ((a.astype(np.uint64) + b) % P).astype(np.uint32))
    wide = be.columns(6, 10)                                   # 3 x 2^10 u64 lanes
    be._chk(be.L.nx_m31_widen(be.ctx, wide.ptr, da.ptr, C.c_size_t(n)))
    w = wide.to_cpu().reshape(-1).view(np.uint64)
    assert np.array_equal(w, a.reshape(-1).astype(np.uint64))
    w8 = (w * 8 + 7).astype(np.uint64)                         # what an 8-rank sum all-reduce could leave in a lane
    be._chk(be.L.nx_upload(be.ctx, wide.ptr, w8.view(np.uint32).ctypes.data_as(C.c_void_p), C.c_size_t(2 * n)))
    be._chk(be.L.nx_m31_narrow(be.ctx, dc.ptr, wide.ptr, C.c_size_t(n)))
    assert np.array_equal(dc.to_cpu().reshape(-1), (w8 % P).astype(np.uint32))


# ---------------- the remaining Backend supertraits: each export against oracle/backend_ops.h -----------------

def test_batch_inverse_matches_oracle(be, oracle):
    L = oracle.lib()
    for log in (0, 5, 13, 18):
        vals = np.random.default_rng(log + 400).integers(1, P, (3, 1 << log), dtype=np.uint32)      # non-zero, as Stwo requires
        got = be.batch_inverse_m31(be.columns_from_host(vals)).to_cpu()
        ref = np.zeros_like(vals)
        for c in range(3):
            L.orc_batch_inverse_m31(O.ptr(np.ascontiguousarray(vals[c])), O.ptr(ref[c]), C.c_size_t(1 << log))
        assert np.array_equal(got, ref), log
        sec = np.random.default_rng(log + 500).integers(0, P, (4, 1 << log), dtype=np.uint32)
        sec[0] |= 1                                                                                   # never the zero element
        got4 = be.batch_inverse_qm31(be.columns_from_host(sec)).to_cpu()
        ref4 = [np.zeros(1 << log, np.uint32) for _ in range(4)]
        L.orc_batch_inverse_qm31(O.ptr_array([np.ascontiguousarray(c) for c in sec]), O.ptr_array(ref4), C.c_size_t(1 << log))
        assert np.array_equal(got4, np.stack(ref4)), log


def test_secure_accumulate_and_powers_match_oracle(be, oracle):
    L = oracle.lib()
    log = 11
    a = rand_cols(1, 4, log); b = rand_cols(2, 4, log)
    got = be.secure_accumulate(be.columns_from_host(a), be.columns_from_host(b)).to_cpu()
    ref = [np.ascontiguousarray(c).copy() for c in a]
    L.orc_secure_accumulate(O.ptr_array(ref), O.ptr_array([np.ascontiguousarray(c) for c in b]), C.c_size_t(1 << log))
    assert np.array_equal(got, np.stack(ref))
    _, _, felt = _random_point_and_alpha(oracle, 33)
    pw = be.generate_secure_powers(felt, 37)
    rp = np.zeros((37, 4), np.uint32)
    L.orc_generate_secure_powers(O.ptr(felt), C.c_size_t(37), O.ptr(rp))
    assert np.array_equal(pw, rp) and list(pw[0]) == [1, 0, 0, 0] and np.array_equal(pw[1], felt)


def test_bit_reverse_secure(be):
    log = 9
    v = rand_cols(3, 4, log)
    got = be.bit_reverse_secure(be.columns_from_host(v)).to_cpu()
    idx = np.array([int(format(i, "0%db" % log)[::-1], 2) for i in range(1 << log)])
    assert np.array_equal(got, v[:, idx])


@pytest.mark.parametrize("mode", [0, 1])
def test_commit_on_layer_matches_oracle_and_whole_tree(be, oracle, mode):
    """MerkleOps::commit_on_layer one layer at a time (leaf layer without a previous layer, inner layers with and without injected
    columns) == the oracle's layer rule == the corresponding layer of nx_merkle_commit."""
    L = oracle.lib()
    be.set_hash_mode(mode)
    try:
        big, small = rand_cols(70, 19, 8), rand_cols(71, 3, 7)
        d_big, d_small = be.columns_from_host(big), be.columns_from_host(small)
        tree = be.merkle_commit([d_big, d_small])
        leaf = be.merkle_commit_on_layer(8, None, d_big)
        ref_leaf = np.zeros(8 << 8, np.uint32)
        L.orc_commit_on_layer(8, None, O.ptr_array([np.ascontiguousarray(c) for c in big]), C.c_size_t(19), mode, O.ptr(ref_leaf))
        assert np.array_equal(leaf.to_cpu().reshape(-1), ref_leaf)
        assert np.array_equal(tree.layer(8).reshape(-1), ref_leaf)
        l7 = be.merkle_commit_on_layer(7, leaf.ptr.value, d_small)
        ref7 = np.zeros(8 << 7, np.uint32)
        L.orc_commit_on_layer(7, O.ptr(ref_leaf), O.ptr_array([np.ascontiguousarray(c) for c in small]), C.c_size_t(3), mode, O.ptr(ref7))
        assert np.array_equal(l7.to_cpu().reshape(-1), ref7) and np.array_equal(tree.layer(7).reshape(-1), ref7)
        l6 = be.merkle_commit_on_layer(6, l7.ptr.value, None)
        ref6 = np.zeros(8 << 6, np.uint32)
        L.orc_commit_on_layer(6, O.ptr(ref7), None, C.c_size_t(0), mode, O.ptr(ref6))
        assert np.array_equal(l6.to_cpu().reshape(-1), ref6) and np.array_equal(tree.layer(6).reshape(-1), ref6)
    finally:
        be.set_hash_mode(0)


@pytest.mark.parametrize("mode", [0, 1])
def test_merkle_decommit_matches_oracle(be, oracle, mode):
    """MerkleProver::decommit on a mixed-degree tree with queries on two layers (the shape of a FRI first layer and of a
    prover2-style trace tree): queried values, hash witness and column witness, word for word."""
    L = oracle.lib()
    be.set_hash_mode(mode)
    try:
        logs = [9] * 5 + [7] * 3 + [9] * 2 + [4]
        host = [np.random.default_rng(900 + i).integers(0, P, 1 << l, dtype=np.uint32) for i, l in enumerate(logs)]
        sets = [be.columns_from_host(h) for h in host]
        tree = be.merkle_commit(sets)
        queries = {9: [3, 4, 200, 201, 511], 7: [0, 50], 4: [7]}
        qv, hw, cw = be.merkle_decommit(tree, sets, queries)
        qlogs = np.array(sorted(queries), np.int32)
        qcnt = np.array([len(queries[int(l)]) for l in qlogs], np.int32)
        qs = np.array([q for l in qlogs for q in queries[int(l)]], np.uint64)
        n_out = (C.c_size_t * 3)()
        r_qv, r_hw, r_cw = np.zeros(4096, np.uint32), np.zeros(8 * 4096, np.uint32), np.zeros(4096, np.uint32)
        L.orc_merkle_decommit(O.ptr_array(host), O.ptr(np.array(logs, np.int32)), len(logs), mode, O.ptr(qlogs), O.ptr(qcnt), len(qlogs), O.ptr(qs),
                              O.ptr(r_qv), O.ptr(r_hw), O.ptr(r_cw), n_out)
        assert len(qv) == n_out[0] and np.array_equal(qv, r_qv[:n_out[0]])
        assert len(hw) == n_out[1] and np.array_equal(hw.reshape(-1), r_hw[:8 * n_out[1]])
        assert len(cw) == n_out[2] and np.array_equal(cw, r_cw[:n_out[2]])
        assert n_out[1] > 0 and n_out[2] > 0
    finally:
        be.set_hash_mode(0)


def test_fri_decompose_matches_oracle(be, oracle):
    L = oracle.lib()
    for log in (1, 4, 12, 17):
        src = rand_cols(log + 600, 4, log)
        g, lam = be.fri_decompose(be.columns_from_host(src))
        ref = [np.zeros(1 << log, np.uint32) for _ in range(4)]
        rl = np.zeros(4, np.uint32)
        L.orc_fri_decompose(O.ptr_array([np.ascontiguousarray(c) for c in src]), log, O.ptr_array(ref), O.ptr(rl))
        assert np.array_equal(lam, rl), log
        assert np.array_equal(g.to_cpu(), np.stack(ref)), log


def test_session_draw_felts_matches_oracle_channel(be, nz, oracle):
    """Channel::draw_felts(n) takes two secure felts from every Blake2s draw; n single draws take one each."""
    L = oracle.lib()
    s = be.prover_session(nz.default_config(), 6)
    s.mix_u64(77)
    got = s.draw_felts(5)
    ch = C.c_void_p(L.orc_channel_new())
    L.orc_channel_mix_u64(ch, 77)
    ref = np.zeros((5, 4), np.uint32)
    L.orc_channel_draw_secure_felts(ch, C.c_size_t(5), O.ptr(ref))
    assert np.array_equal(got, ref)
    nxt, rn = s.draw_felt(), np.zeros(4, np.uint32)
    L.orc_channel_draw_secure_felt(ch, O.ptr(rn))
    assert np.array_equal(nxt, rn)
    L.orc_channel_free(ch)
    s.close()


def test_context_on_a_worker_thread_uses_its_own_device(nz):
    """every entry makes the context's device current.  A context created on the main thread is driven from a fresh worker
    thread (whose current device is whatever HIP defaults to) — with more than one GPU visible the context lives on the LAST device."""
    import threading, torch
    dev = torch.cuda.device_count() - 1
    b = nz.HipBackend(dev)
    vals = rand_cols(5, 3, 14)
    ref_be = nz.HipBackend(0)
    tw0 = ref_be.precompute_twiddles(14)
    expect = ref_be.lde(tw0, ref_be.columns_from_host(vals), 1).to_cpu()
    out, errs = [], []

    def worker():
        try:
            tw = b.precompute_twiddles(14)
            cols = b.columns_from_host(vals)
            out.append(b.lde(tw, cols, 1).to_cpu())
            out.append(b.merkle_commit([cols]).root())
        except Exception as e:
            errs.append(e)
    t = threading.Thread(target=worker); t.start(); t.join()
    assert not errs, errs
    assert np.array_equal(out[0], expect)
    b.close(); ref_be.close()
