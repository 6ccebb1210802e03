"""nx_proof_serialize_stwo (host-only entry of libnexus_hip.so): the postcard bytes of the reference's
`Proof { stark_proof: StarkProof<Blake2sMerkleHasher>, claimed_sum, log_size }` (reference prover/src/machine.rs:93-98,
sdk/Cargo.toml:22) from an NXP1 proof.  An independent decoder — written here from the postcard wire rules and the struct
declarations, not from the encoder — must reproduce every field of the NXP1 stream; malformed input is refused.
[The field ORDER is upstream-recollection of Stwo's derive(Serialize); tools/dump_reference.rs pins it on a box with cargo.]"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O


class Post:
    def __init__(self, b):
        self.b, self.i = b, 0

    def varint(self):
        v, s = 0, 0
        while True:
            x = self.b[self.i]; self.i += 1
            v |= (x & 0x7F) << s; s += 7
            if x < 0x80:
                return v

    def hash(self):
        h = self.b[self.i:self.i + 32]; self.i += 32
        return list(np.frombuffer(bytes(h), np.uint32))

    def q(self):
        return [self.varint() for _ in range(4)]

    def vec(self, f):
        return [f() for _ in range(self.varint())]

    def decommit(self):
        return {"hash_witness": self.vec(self.hash), "column_witness": self.vec(self.varint)}

    def fri_layer(self):
        return {"fri_witness": self.vec(self.q), "decommitment": self.decommit(), "commitment": self.hash()}


def decode(b):
    p = Post(b)
    out = {"pow_bits": p.varint(), "log_blowup": p.varint(), "log_last": p.varint(), "n_queries": p.varint()}
    out["commitments"] = p.vec(p.hash)
    out["sampled_values"] = p.vec(lambda: p.vec(lambda: p.vec(p.q)))
    out["decommitments"] = p.vec(p.decommit)
    out["queried_values"] = p.vec(lambda: p.vec(p.varint))
    out["pow"] = p.varint()
    out["first_layer"] = p.fri_layer()
    out["inner_layers"] = p.vec(p.fri_layer)
    out["last_coeffs"] = p.vec(p.q)
    out["last_log"] = p.varint()
    out["claimed_sum"] = p.vec(p.q)
    out["log_size"] = p.vec(p.varint)
    assert p.i == len(b)
    return out


def reencode_nxp1(d):
    """The NXP1 word stream the decoded fields correspond to (oracle/pcs.h proof_serialize layout)."""
    w = [0x3150584E, d["pow_bits"], d["log_blowup"], d["n_queries"], d["log_last"], len(d["commitments"])]
    for h in d["commitments"]:
        w += h
    for t in d["sampled_values"]:
        w.append(len(t))
        for c in t:
            w.append(len(c))
            for q in c:
                w += q

    def dec(x):
        r = [len(x["hash_witness"])]
        for h in x["hash_witness"]:
            r += h
        return r + [len(x["column_witness"])] + x["column_witness"]

    def layer(x):
        r = [len(x["fri_witness"])]
        for q in x["fri_witness"]:
            r += q
        return r + dec(x["decommitment"]) + x["commitment"]
    for x in d["decommitments"]:
        w += dec(x)
    for v in d["queried_values"]:
        w += [len(v)] + v
    w += [d["pow"] & 0xFFFFFFFF, d["pow"] >> 32]
    w += layer(d["first_layer"]) + [len(d["inner_layers"])]
    for x in d["inner_layers"]:
        w += layer(x)
    n = len(d["last_coeffs"]); lg = d["last_log"]
    assert n == 1 << lg
    rev = lambda i: int(format(i, "0%db" % lg)[::-1], 2) if lg else 0
    w.append(n)
    for k in range(n):
        w += d["last_coeffs"][rev(k)]                 # LinePoly stores bit-reversed coefficients; NXP1 the ordered ones
    return np.array(w, np.uint32)


def serialize(words, claimed, logs):
    import nexus_zkvm_amd as nz
    L = nz.load_library()
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t(0)
    cs, lg = np.ascontiguousarray(claimed, np.uint32).reshape(-1), np.ascontiguousarray(logs, np.uint32)
    w = np.ascontiguousarray(words, np.uint32)
    rc = L.nx_proof_serialize_stwo(w.ctypes.data_as(C.c_void_p), C.c_size_t(len(w)), cs.ctypes.data_as(C.c_void_p), lg.ctypes.data_as(C.c_void_p), len(lg), C.byref(out), C.byref(n))
    if rc != 0:
        return None
    b = bytes(bytearray(out[:n.value]))
    L.nx_free_host(out)
    return b


@pytest.mark.parametrize("comps,kw", [([(6, 3, 9, 4)], dict(pow_bits=4)), ([(7, 2, 20, 8), (5, 2, 4, 0)], dict(pow_bits=3, log_last_layer_degree_bound=2))])
def test_postcard_proof_round_trips_through_an_independent_decoder(oracle, comps, kw):
    cfg = O.default_cfg(pow_bits=kw["pow_bits"], log_last=kw.get("log_last_layer_degree_bound", 0))
    words = O.prove_synth(comps, cfg, seed=3)
    claimed = np.array([[k + 1, 2, 3, 2147483646] for k in range(len(comps))], np.uint32)
    b = serialize(words, claimed, [c[0] for c in comps])
    assert b is not None
    d = decode(b)
    assert np.array_equal(reencode_nxp1(d), words)
    assert d["claimed_sum"] == claimed.tolist() and d["log_size"] == [c[0] for c in comps]
    assert d["n_queries"] == 3 and d["log_blowup"] == 1 and len(d["commitments"]) == 4
    # hashes travel as 32 raw bytes, field elements as varints (up to 5 bytes for a 31-bit value)
    assert len(b) <= 5 * len(words) + 64


def test_malformed_proofs_are_refused(oracle):
    words = O.prove_synth([(5, 2, 4, 0)], O.default_cfg(pow_bits=2), seed=1)
    assert serialize(words[:-3], np.zeros((1, 4)), [5]) is None
    bad = words.copy(); bad[0] ^= 1
    assert serialize(bad, np.zeros((1, 4)), [5]) is None
    longer = np.concatenate([words, np.zeros(2, np.uint32)])
    assert serialize(longer, np.zeros((1, 4)), [5]) is None
