"""Repeats the same prove and checks the proof bytes never change (races between the FFT streams / hash stream would show here)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nexus_zkvm_amd as nz
be = nz.HipBackend(0)
for comps in ([(22, 27, 347, 64)], [(18, 27, 347, 64), (16, 4, 40, 8), (13, 2, 7, 4)], [(20, 8, 101, 33)]):
    hs = set()
    for i in range(6):
        w = be.prove(comps, nz.default_config(), seed=99)
        hs.add(hashlib.sha256(w.tobytes()).hexdigest())
    print(comps, "distinct proofs:", len(hs))
    assert len(hs) == 1
print("deterministic")
