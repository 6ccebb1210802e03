"""Soak test: many proves of several shapes back to back in one context (hand-written path and the recorded-AIR session): the proof
bytes must never change (stream races would show) and the device memory in use must not grow (leaks in the pooled allocator, trees,
FRI layers, JIT kernels would show).  Prints one line per shape."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                                   # before the library (one HIP runtime per process); used for mem_get_info only
import numpy as np
import nexus_zkvm_amd as nz
import nexus_zkvm_amd.air_program as ap
from test_air_program_cpu import synthetic_program

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
be = nz.HipBackend(0)
cfg = nz.default_config()


def used_gb():
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 2**30


def session_prove(log, n_pre, n_main, n_inter, comp, kern):
    carr = be._comps([(log, n_pre, n_main, n_inter)])
    s = be.prover_session(cfg, log)

    def fill(tree, n, inter_seed=0):
        ptrs = s.tree_begin([log] * n)
        be._chk(be.L.nx_synth_fill_tree(be.ctx, carr, 1, tree, C.c_uint64(99), C.c_uint64(inter_seed), (C.c_void_p * max(1, n))(*ptrs)))
        s.tree_commit()
    s.mix_u64(log)
    fill(0, n_pre); fill(1, n_main)
    z = s.draw_felt()
    s.mix_felts(np.zeros(4, np.uint32))
    fill(2, n_inter, (int(z[0]) << 32) ^ int(z[1]) ^ (int(z[2]) << 16) ^ (int(z[3]) << 48))
    w = s.prove([comp], kernels=[kern])
    s.close()
    return w


for comps in ([(22, 27, 347, 64)], [(18, 27, 347, 64), (16, 4, 40, 8), (13, 2, 7, 4)], [(20, 8, 101, 33)], [(12, 3, 20, 19)]):
    hs, mem = set(), []
    one = len(comps) == 1
    if one:
        log, a, b, c = comps[0]
        cols = [(0, k) for k in range(a)] + [(1, k) for k in range(b)] + [(2, k) for k in range(c)]
        comp = ap.Component(log, synthetic_program(ap, a, b, c), cols)
        kern = be.compile_air(comp.program, len(cols))
    t0 = time.perf_counter()
    hm = set()
    for i in range(reps):
        hs.add(hashlib.sha256(be.prove(comps, cfg, seed=99).tobytes()).hexdigest())
        hm.add(hashlib.sha256(be.prove_machine([(lg, x, y, 4 * (z // 4)) for lg, x, y, z in comps], cfg, seed=99).tobytes()).hexdigest())   # real logup, recorded AIR
        if one:
            hs.add(hashlib.sha256(session_prove(log, a, b, c, comp, kern).tobytes()).hexdigest())
        if i in (2, reps - 1):
            be.sync(); mem.append(used_gb())
    print(comps, "proves:", reps * (3 if one else 2), "distinct proofs:", len(hs), "+", len(hm), "(machine)", "device memory in use after 3 / after all (GiB): %.2f / %.2f" % (mem[0], mem[1]),
          "%.1f ms per prove" % ((time.perf_counter() - t0) * 1e3 / (reps * (3 if one else 2))))
    assert len(hs) == 1 and len(hm) == 1 and mem[1] <= mem[0] + 0.25
print("soak ok")
