"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

P = (1 << 31) - 1
HASH_STD, HASH_RAW0 = 0, 1
FRI_ALPHA_PREV, FRI_ALPHA_FIRST = 0, 1

u32p = C.POINTER(C.c_uint32)
u32pp = C.POINTER(u32p)


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".h", ".cpp"))]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build_oracle()
        L = C.CDLL(LIB_PATH)
        L.orc_m31_mul.restype = C.c_uint32
        L.orc_m31_add.restype = C.c_uint32
        L.orc_m31_sub.restype = C.c_uint32
        L.orc_m31_inv.restype = C.c_uint32
        L.orc_m31_reduce.restype = C.c_uint32
        L.orc_m31_reduce.argtypes = [C.c_uint64]
        L.orc_bit_reverse_index.restype = C.c_uint32
        L.orc_coset_index_to_circle_domain_index.restype = C.c_uint32
        L.orc_eval_basis_at_m31_point.restype = C.c_uint32
        L.orc_twiddles_new.restype = C.c_void_p
        L.orc_twiddles_free.argtypes = [C.c_void_p]
        L.orc_channel_new.restype = C.c_void_p
        L.orc_channel_grind.restype = C.c_uint64
        L.orc_channel_mix_u64.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_channel_verify_pow.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
        L.orc_prove_synth.restype = u32p
        L.orc_prove_synth.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t, C.c_int,
                                      C.POINTER(C.c_size_t)]
        L.orc_verify_synth.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.orc_last_error.restype = C.c_char_p
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_time_prove_synth.restype = C.c_double
        L.orc_time_prove_synth.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_int]
        L.orc_inter_seed_from.restype = C.c_uint64
        L.orc_synth_tree_columns.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        L.orc_synth_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_blake2s.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        for f in ("orc_prover_new", "orc_prover_channel", "orc_verifier_new", "orc_verifier_channel"):
            getattr(L, f).restype = C.c_void_p
        L.orc_prover_new.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_verifier_new.argtypes = [C.c_void_p]
        for f in ("orc_prover_free", "orc_prover_channel", "orc_verifier_free", "orc_verifier_channel"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_prover_commit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_prover_prove.restype = u32p
        L.orc_prover_prove.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_verifier_commit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_verifier_verify.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def ptr_array(arrs):
    """uint32_t*[] from a list of contiguous uint32 numpy arrays."""
    arr = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    return arr


def u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def default_cfg(pow_bits=10, log_blowup=1, n_queries=3, log_last=0, hash_mode=HASH_STD, fri_alpha_mode=FRI_ALPHA_PREV,
                log_constraint_degree=1):
    return np.array([pow_bits, log_blowup, n_queries, log_last, hash_mode, fri_alpha_mode, log_constraint_degree],
                    dtype=np.int32)


def comps_array(comps):
    """comps: list of (log_size, n_pre, n_main, n_inter[, log_constraint_degree_bound]) — the bound defaults to 0 = the config's."""
    return np.array([(tuple(c) + (0,) * 5)[:5] for c in comps], dtype=np.int32).reshape(-1, 5).copy()   # a 6th entry (the machine's logup mode) is the checker's (tests/machine_ref.py), not the oracle's


class Twiddles:
    def __init__(self, root_log):
        self.root_log = root_log
        self.h = C.c_void_p(lib().orc_twiddles_new(root_log))

    def arrays(self):
        n = 1 << self.root_log
        tw, itw = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        lib().orc_twiddles_get(self.h, ptr(tw), ptr(itw))
        return tw, itw

    def interpolate(self, values):
        v = u32(values).copy()
        lib().orc_interpolate(self.h, ptr(v), int(np.log2(len(v))))
        return v

    def evaluate(self, coeffs, log_out):
        c = u32(coeffs)
        out = np.zeros(1 << log_out, np.uint32)
        lib().orc_evaluate(self.h, ptr(c), int(np.log2(len(c))), ptr(out), log_out)
        return out

    def __del__(self):
        try:
            lib().orc_twiddles_free(self.h)
        except Exception:
            pass


def finalize_column(nat):
    nat = u32(nat)
    out = np.zeros_like(nat)
    lib().orc_finalize_column(ptr(nat), ptr(out), int(np.log2(len(nat))))
    return out


def eval_at_point(coeffs, pt8):
    c = u32(coeffs)
    out = np.zeros(4, np.uint32)
    p = u32(pt8)
    lib().orc_eval_at_point(ptr(c), int(np.log2(len(c))), ptr(p), ptr(out))
    return out


def merkle_commit(cols, mode=HASH_STD, want_layers=False):
    cols = [u32(c) for c in cols]
    logs = np.array([int(np.log2(len(c))) for c in cols], np.int32)
    root = np.zeros(8, np.uint32)
    layers = None
    if want_layers:
        mx = int(logs.max()) if len(cols) else 0
        layers = np.zeros(8 * ((2 << mx) - 1), np.uint32)
    lib().orc_merkle_commit(ptr_array(cols), ptr(logs), len(cols), mode, ptr(root), ptr(layers) if want_layers else None)
    return (root, layers) if want_layers else root


def lde_commit(cols, log_blowup=1, mode=HASH_STD, threads=4, root_log=None):
    """Reference of nx_lde_commit (config #2): interpolate + evaluate every column (in place: `cols` become the coefficients),
    Blake2s Merkle root of the LDE columns.  Returns (lde columns, root)."""
    logs = np.array([int(np.log2(len(c))) for c in cols], np.int32)
    tw = Twiddles(int(logs.max()) + log_blowup if root_log is None else root_log)
    ldes = [np.zeros(len(c) << log_blowup, np.uint32) for c in cols]
    root = np.zeros(8, np.uint32)
    lib().orc_lde_commit(tw.h, ptr_array(cols), ptr(logs), len(cols), log_blowup, mode, threads, ptr_array(ldes), ptr(root))
    return ldes, root


def synth_tree_columns(comps, tree, seed, inter_seed=0, threads=4):
    comps = comps_array(comps)
    outs = []
    for (ls, a, b, c, _) in comps:
        n = [a, b, c][tree]
        outs += [np.zeros(1 << ls, np.uint32) for _ in range(n)]
    lib().orc_synth_tree_columns(ptr(comps), len(comps), tree, seed, inter_seed, threads, ptr_array(outs))
    return outs


def prove_synth(comps, cfg, seed=1, ad=b"", threads=4):
    comps = comps_array(comps)
    n = C.c_size_t(0)
    adb = (C.c_uint8 * max(1, len(ad)))(*ad)
    p = lib().orc_prove_synth(ptr(comps), len(comps), ptr(cfg), seed, adb, len(ad), threads, C.byref(n))
    if not p:
        raise RuntimeError("oracle prove failed: " + lib().orc_last_error().decode())
    words = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
    lib().orc_free(p)
    return words


def verify_synth(comps, cfg, words, ad=b""):
    comps = comps_array(comps)
    words = u32(words)
    adb = (C.c_uint8 * max(1, len(ad)))(*ad)
    rc = lib().orc_verify_synth(ptr(comps), len(comps), ptr(cfg), ptr(words), len(words), adb, len(ad))
    return None if rc == 0 else lib().orc_last_error().decode()


def time_prove_synth(comps, cfg, seed=1, threads=1):
    comps = comps_array(comps)
    return lib().orc_time_prove_synth(ptr(comps), len(comps), ptr(cfg), seed, threads)


def eval_constraint_program(program, cols, alpha_powers, denom_inv, log_size, log_eval, acc4=None):
    """oracle/constraints.h: the recorded-constraint program over every row of the evaluation domain; returns the 4
    accumulator columns (numpy uint32, 2^log_eval each)."""
    L = lib()
    cols = [u32(c) for c in cols]
    n = 1 << log_eval
    acc = [np.zeros(n, np.uint32) for _ in range(4)] if acc4 is None else [u32(a).copy() for a in acc4]
    ins = u32(program.instrs).reshape(-1)
    ec = u32(program.econsts).reshape(-1) if len(program.econsts) else np.zeros(4, np.uint32)
    pw = u32(alpha_powers).reshape(-1)
    den = u32(denom_inv)
    L.orc_eval_constraint_program(ptr(ins), len(ins) // 4, program.n_regs, ptr_array(cols), ptr(ec), ptr(pw), ptr(den), log_size, log_eval, ptr_array(acc))
    return acc


def logup_program(program, cols, log_size, n_logup_cols):
    """The interaction trace from a fraction program (oracle/constraints.h logup_program): cols = the component's columns (None for
    columns the program does not load); returns n_logup_cols lists of 4 coordinate columns."""
    ins = np.asarray(program.instrs, dtype=np.uint32).reshape(-1).copy()
    ec = np.asarray(program.econsts, dtype=np.uint32).reshape(-1).copy() if len(program.econsts) else np.zeros(4, np.uint32)
    keep = [u32(c) if c is not None else None for c in cols]
    arr = (C.c_void_p * max(1, len(keep)))(*[c.ctypes.data if c is not None else None for c in keep])
    out = [np.zeros(1 << log_size, np.uint32) for _ in range(4 * n_logup_cols)]
    lib().orc_logup_program(ptr(ins), len(ins) // 4, program.n_regs, arr, ptr(ec), log_size, n_logup_cols, ptr_array(out))
    return [out[4 * j:4 * j + 4] for j in range(n_logup_cols)]


def qm31_mul(a, b):
    a, b, o = u32(a), u32(b), np.zeros(4, np.uint32)
    lib().orc_qm31_mul(ptr(a), ptr(b), ptr(o))
    return o


def logup_combine(cols, alpha_powers, z):
    cols = [u32(c) for c in cols]
    log = int(np.log2(len(cols[0])))
    out = [np.zeros(1 << log, np.uint32) for _ in range(4)]
    lib().orc_logup_combine(ptr_array(cols), len(cols), ptr(u32(alpha_powers).reshape(-1)), ptr(u32(z)), log, ptr_array(out))
    return out


def logup_finalize_col(den_a, scale_a=(1, 0, 0, 0), mult_a=None, den_b=None, scale_b=(1, 0, 0, 0), mult_b=None, prev=None):
    den_a = [u32(c) for c in den_a]
    log = int(np.log2(len(den_a[0])))
    out = [np.zeros(1 << log, np.uint32) for _ in range(4)]
    keep = [u32(mult_a) if mult_a is not None else None, u32(mult_b) if mult_b is not None else None,
            [u32(c) for c in den_b] if den_b is not None else None, [u32(c) for c in prev] if prev is not None else None]
    lib().orc_logup_finalize_col(log, ptr(keep[0]) if keep[0] is not None else None, ptr(u32(scale_a)), ptr_array(den_a),
                                 ptr(keep[1]) if keep[1] is not None else None, ptr(u32(scale_b)), ptr_array(keep[2]) if keep[2] is not None else None,
                                 ptr_array(keep[3]) if keep[3] is not None else None, ptr_array(out))
    return out


def logup_finalize_last(col4):
    col = [u32(c).copy() for c in col4]
    log = int(np.log2(len(col[0])))
    cs = np.zeros(4, np.uint32)
    lib().orc_logup_finalize_last(log, ptr_array(col), ptr(cs))
    return col, cs


class _SessionChannel:
    """The Blake2sChannel of a prover / verifier session (owned by the session)."""

    def __init__(self, h):
        self.h = C.c_void_p(h)

    def mix_u64(self, v):
        lib().orc_channel_mix_u64(self.h, v)

    def mix_felts(self, felts):
        f = u32(felts).reshape(-1)
        lib().orc_channel_mix_felts(self.h, ptr(f), C.c_size_t(len(f) // 4))

    def draw_felt(self):
        out = np.zeros(4, np.uint32)
        lib().orc_channel_draw_secure_felt(self.h, ptr(out))
        return out

    def draw_felts(self, n):
        out = np.zeros((n, 4), np.uint32)
        lib().orc_channel_draw_secure_felts(self.h, C.c_size_t(n), ptr(out))
        return out

    def digest(self):
        out = np.zeros(8, np.uint32)
        lib().orc_channel_digest(self.h, ptr(out))
        return out


class ProverSession(_SessionChannel):
    """oracle/air_generic.h::ProverSession — CommitmentSchemeProver + channel + stwo::prover::prove over recorded AIRs."""

    def __init__(self, cfg, max_log, threads=4):
        self.cfg = cfg
        self.p = C.c_void_p(lib().orc_prover_new(ptr(cfg), max_log, threads))
        super().__init__(lib().orc_prover_channel(self.p))

    def commit(self, cols):
        cols = [u32(c) for c in cols]
        logs = np.array([int(np.log2(len(c))) for c in cols], np.int32)
        root = np.zeros(8, np.uint32)
        lib().orc_prover_commit(self.p, ptr_array(cols) if cols else None, ptr(logs), len(cols), ptr(root))
        return root

    def prove(self, components):
        w = encode_air(components)
        n = C.c_size_t(0)
        p = lib().orc_prover_prove(self.p, ptr(w), len(w), C.byref(n))
        if not p:
            raise RuntimeError("oracle prove failed: " + lib().orc_last_error().decode())
        words = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
        lib().orc_free(p)
        return words

    def __del__(self):
        try:
            lib().orc_prover_free(self.p)
        except Exception:
            pass


class VerifierSession(_SessionChannel):
    """oracle/air_generic.h::VerifierSession — CommitmentSchemeVerifier + channel + core::verifier::verify."""

    def __init__(self, cfg):
        self.v = C.c_void_p(lib().orc_verifier_new(ptr(cfg)))
        super().__init__(lib().orc_verifier_channel(self.v))

    def commit(self, root, logs):
        r, l = u32(root), np.array(logs, np.int32)
        lib().orc_verifier_commit(self.v, ptr(r), ptr(l), len(l))

    def verify(self, components, words):
        """None when the proof is accepted, else the verifier's error text."""
        w, pw = encode_air(components), u32(words)
        rc = lib().orc_verifier_verify(self.v, ptr(w), len(w), ptr(pw), len(pw))
        return None if rc == 0 else lib().orc_last_error().decode()

    def __del__(self):
        try:
            lib().orc_verifier_free(self.v)
        except Exception:
            pass


def proof_header_words():
    """NXP1 (oracle/pcs.h::proof_serialize): magic, pow_bits, log_blowup, n_queries, log_last_layer_degree_bound; then the
    commitment count and the roots."""
    return 5


def encode_component(c):
    """The flat u32 description of a nexus_zkvm_amd.air_program.Component that oracle/air_generic.h::gair_decode reads."""
    pr = c.program
    ins = np.asarray(pr.instrs, dtype=np.uint32).reshape(-1)
    ec = np.asarray(pr.econsts, dtype=np.uint32).reshape(-1)
    offs = [o for m in c.masks for o in m]
    head = [c.log_size, len(ins) // 4, pr.n_regs, len(ec) // 4, pr.n_constraints, len(c.cols), len(offs), getattr(c, "log_constraint_degree_bound", 0)]
    parts = [np.array(head, np.uint32), ins, ec, np.array([t for t, _ in c.cols], np.uint32), np.array([i for _, i in c.cols], np.uint32),
             np.array([len(m) for m in c.masks], np.uint32), np.array(offs, np.int32).view(np.uint32)]
    return np.concatenate(parts)


def encode_air(components):
    return np.concatenate([np.array([len(components)], np.uint32)] + [encode_component(c) for c in components])
