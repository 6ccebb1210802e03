"""Host-only entry points of libnexus_hip.so against malformed input (no GPU): the proof re-encoder (nx_proof_serialize_stwo) and the
recorded-AIR validator / source generator (nx_air_compile_subset with a NULL context, nx_air_constraint_degrees), the fraction-program
validator / generator (nx_logup_program with a NULL context) and nx_machine_air_program.  Each fuzzer runs in a
child process — a crash is a failed test, not a dead test session — and must finish with errors returned, never with a signal."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SERDE = r'''
import sys, ctypes as C
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import oracle_lib as O
import nexus_zkvm_amd as nz
O.build_oracle()
L = nz.load_library()
comps = [(7, 2, 20, 8), (5, 2, 4, 0)]
words = np.array(O.prove_synth(comps, O.default_cfg(pow_bits=3, log_last=2), seed=3), np.uint32)
claimed = np.arange(8, dtype=np.uint32); logs = np.array([7, 5], np.uint32)
def ser(w):
    w = np.ascontiguousarray(w, np.uint32)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t(0)
    rc = L.nx_proof_serialize_stwo(w.ctypes.data_as(C.c_void_p), C.c_size_t(len(w)), claimed.ctypes.data_as(C.c_void_p), logs.ctypes.data_as(C.c_void_p), 2, C.byref(out), C.byref(n))
    if rc == 0: L.nx_free_host(out)
    return rc
assert ser(words) == 0
refused = sum(ser(words[:k]) != 0 for k in range(len(words)))
assert refused == len(words), "a truncated proof was accepted"
rng = np.random.default_rng(1)
for it in range(3000):
    w = words.copy()
    for _ in range(int(rng.integers(1, 4))):
        i = int(rng.integers(0, len(w)))
        w[i] = [0xFFFFFFFF, 0x7FFFFFFF, 0, 1 << 30, int(rng.integers(0, 1 << 32)), (int(w[i]) + 1) & 0xFFFFFFFF, 1 << 20][int(rng.integers(0, 7))]
    ser(w)
print("done")
'''

AIR = r'''
import sys, ctypes as C
sys.path.insert(0, %(root)r)
import numpy as np
import nexus_zkvm_amd as nz
L = nz.load_library()
rng = np.random.default_rng(7)
ok = 0
for it in range(4000):
    n = int(rng.integers(0, 40))
    prog = np.zeros((n, 4), np.uint32)
    wild = rng.random() < 0.5
    for k in range(n):
        prog[k, 0] = int(rng.integers(0, 15)) if rng.random() < 0.95 else int(rng.integers(0, 1 << 32))
        for j in (1, 2, 3):
            prog[k, j] = int(rng.integers(0, 1 << 32)) if (wild and rng.random() < 0.2) else int(rng.integers(0, 12))
    n_regs = int(rng.integers(0, 16)) if rng.random() < 0.9 else int(rng.integers(0, 1 << 32))
    n_cols = int(rng.integers(0, 12)); n_ec = int(rng.integers(0, 4))
    n_c = int((prog[:, 0] == 13).sum() + (prog[:, 0] == 14).sum()) if rng.random() < 0.8 else int(rng.integers(0, 1 << 16))
    p = np.ascontiguousarray(prog.reshape(-1), np.uint32)
    src = C.c_char_p()
    if L.nx_air_compile_subset(None, p.ctypes.data_as(C.c_void_p), n, n_regs, n_cols, n_ec, n_c, None, None, C.byref(src)) == 0:
        ok += 1; L.nx_free_host(src)
    deg = (C.c_uint32 * max(1, min(n_c, 1 << 16)))()
    L.nx_air_constraint_degrees(None, p.ctypes.data_as(C.c_void_p), n, n_regs, n_cols, n_ec, n_c, deg)
assert ok > 0, "no random program was valid: the fuzzer does not reach the generator"
print("done")
'''


LOGUP = r'''
import sys, ctypes as C
sys.path.insert(0, %(root)r)
import numpy as np
import nexus_zkvm_amd as nz
L = nz.load_library()
rng = np.random.default_rng(11)
ok = 0
for it in range(4000):
    n = int(rng.integers(0, 40))
    prog = np.zeros((n, 4), np.uint32)
    wild = rng.random() < 0.5
    n_lc = int(rng.integers(0, 5))
    batch = 0
    for k in range(n):
        r = rng.random()
        prog[k, 0] = int(rng.integers(0, 13)) if r < 0.7 else int(rng.integers(15, 17)) if r < 0.95 else int(rng.integers(0, 1 << 32))
        for j in (1, 2, 3):
            prog[k, j] = int(rng.integers(0, 1 << 32)) if (wild and rng.random() < 0.2) else int(rng.integers(0, 12))
        if prog[k, 0] in (15, 16) and rng.random() < 0.9:          # mostly well-ordered batches, so that valid programs occur
            prog[k, 1] = batch
            if rng.random() < 0.5: batch += 1
    n_regs = int(rng.integers(0, 20)) if rng.random() < 0.9 else int(rng.integers(0, 1 << 32))
    n_cols = int(rng.integers(0, 12)); n_ec = int(rng.integers(0, 4))
    p = np.ascontiguousarray(prog.reshape(-1), np.uint32)
    ec = np.zeros(max(1, 4 * n_ec), np.uint32)
    src = C.c_char_p()
    if L.nx_logup_program(None, p.ctypes.data_as(C.c_void_p), n, n_regs, None, n_cols, ec.ctypes.data_as(C.c_void_p), n_ec, int(rng.integers(0, 33)), n_lc, None, C.byref(src)) == 0:
        ok += 1; L.nx_free_host(src)
    # the machine's recorded program for random shapes and logup modes: refused or handed out, never a crash
    spec = nz.ComponentSpec(int(rng.integers(0, 30)), int(rng.integers(0, 6)), int(rng.integers(0, 40)), int(rng.integers(0, 40)), int(rng.integers(0, 3)), int(rng.integers(0, 9)))
    pp, a, b, c = C.c_void_p(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    if L.nx_machine_air_program(C.byref(spec), int(rng.integers(0, 4)), C.byref(pp), C.byref(a), C.byref(b), C.byref(c)) == 0:
        L.nx_free_host(pp)
assert ok > 0, "no random fraction program was valid: the fuzzer does not reach the generator"
print("done")
'''


def _run(body):
    r = subprocess.run([sys.executable, "-c", body % {"root": ROOT}], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("done"), (r.returncode, r.stdout[-500:], r.stderr[-1500:])


def test_proof_reencoder_survives_truncated_and_mutated_proofs():
    _run(SERDE)


def test_air_validator_and_generator_survive_random_programs():
    _run(AIR)


def test_fraction_program_validator_and_generator_survive_random_programs():
    """nx_logup_program's host half (validation of batch order / registers / columns, source generation) and nx_machine_air_program on random
    input: errors, never a signal."""
    _run(LOGUP)
