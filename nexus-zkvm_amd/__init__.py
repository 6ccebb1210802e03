"""nexus_zkvm_amd — MI355X (gfx950) backend for the Nexus zkVM commit-and-prove hot path.

Thin ctypes mirror of include/nexus_hip.h.  The names follow the Stwo backend traits the reference
instantiates on `SimdBackend` (reference prover/src/machine.rs:16,186,203,286):
`HipBackend.precompute_twiddles`, `.interpolate_columns`, `.evaluate_polynomials`, `.commit` (Merkle),
`.eval_at_points`, `.accumulate_quotients`, `.fold_line`, `.fold_circle_into_line`, `.grind`, and the
synthetic-machine `prove`.  There is NO CPU fallback: importing works without a GPU (so the symbol
table can be checked), but creating a `HipBackend` raises if libnexus_hip.so or a gfx950 device is
missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NX_LIB") or os.path.join(_HERE, "libnexus_hip.so")   # NX_LIB: A/B builds for tools/ only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "nexus_hip.h")

P = (1 << 31) - 1
NX_OK = 0
HASH_BLAKE2S, HASH_BLAKE2S_RAW0 = 0, 1
FRI_ALPHA_PREV, FRI_ALPHA_FIRST = 0, 1


class LocalGroup:
    """nx_comm_group: the shared board of the in-process transport (one per proof; HipBackend.local_comm(group, rank) per thread)."""

    def __init__(self, world):
        self.world = int(world)
        self.h = C.c_void_p()
        L = load_library()
        rc = L.nx_comm_group_create(self.world, C.byref(self.h))
        if rc != 0:
            raise NexusHipError(f"nx_comm_group_create failed ({rc}): {L.nx_last_error(None).decode()}")

    def reset(self):
        """Re-arm a group a failure broke (nx_comm_group_reset): only when every rank's thread has left its prove call."""
        L = load_library()
        rc = L.nx_comm_group_reset(self.h)
        if rc != 0:
            raise NexusHipError(f"nx_comm_group_reset failed ({rc}): {L.nx_last_error(None).decode()}")

    def broken(self):
        return bool(load_library().nx_comm_group_broken(self.h))

    def peer_access(self, from_rank, to_rank):
        """1 peer to peer / same device, 0 staged by the runtime, -1 a communicator does not exist yet (nx_comm_group_peer_access)"""
        return int(load_library().nx_comm_group_peer_access(self.h, int(from_rank), int(to_rank)))

    def close(self):
        if self.h:
            load_library().nx_comm_group_destroy(self.h)
            self.h = C.c_void_p()


class NexusHipError(RuntimeError):
    pass


class ComponentSpec(C.Structure):
    _fields_ = [("log_size", C.c_uint32), ("n_pre", C.c_uint32), ("n_main", C.c_uint32), ("n_inter", C.c_uint32),
                ("log_constraint_degree_bound", C.c_uint32),   # 0 = the config's log_constraint_degree
                ("logup_mode", C.c_uint32)]                    # nx_prove_machine: LOGUP_PAIRS | LOGUP_ODD | LOGUP_TABLE (0: one fraction per column)


LOGUP_PAIRS, LOGUP_ODD, LOGUP_TABLE = 1, 2, 4                   # nx_component_spec.logup_mode (include/nexus_hip.h NX_LOGUP_*)
TUPLES_V1, TUPLES_KECCAK, TUPLES_V2 = 1 << 4, 2 << 4, 3 << 4    # | NX_LOGUP_TUPLES(k): the reference's relation widths / entry kinds (NX_TUPLES_*)


class PcsConfig(C.Structure):
    _fields_ = [("pow_bits", C.c_uint32), ("log_blowup", C.c_uint32), ("n_queries", C.c_uint32),
                ("log_last_layer_degree_bound", C.c_uint32), ("hash_mode", C.c_uint32), ("fri_alpha_mode", C.c_uint32),
                ("log_constraint_degree", C.c_uint32)]


_SEND_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t)
_RECV_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t)
_ALLREDUCE_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
_ALLGATHER_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_BROADCAST_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32)
_ALLTOALLV_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t))
_ALLGATHER_DEV_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_ABORT_T = C.CFUNCTYPE(None, C.c_void_p)


class NxComm(C.Structure):
    """nx_comm of include/nexus_hip.h: the transport callbacks of one proof on several GPUs."""
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("user", C.c_void_p), ("send", _SEND_T), ("recv", _RECV_T),
                ("allreduce_m31", _ALLREDUCE_T), ("allgather", _ALLGATHER_T), ("broadcast", _BROADCAST_T),
                ("alltoallv", _ALLTOALLV_T), ("allgather_dev", _ALLGATHER_DEV_T), ("abort", _ABORT_T)]


def make_comm(rank, world, impl):
    """Wrap a Python object with send(dst, ptr, n_words) / recv(src, ptr, n_words) / allreduce_m31(ptr, n_words) /
    allgather(bytes) -> list of bytes / broadcast(bytes_or_None, n, root) -> bytes /
    alltoallv(send_ptr, send_off, send_cnt, recv_ptr, recv_off, recv_cnt) (lists of `world` word counts) /
    allgather_dev(send_ptr, n_words, recv_ptr) into an NxComm.  Exceptions become a non-zero return code (the library then fails
    the prove with NX_ERR_HIP)."""
    def guard(f):
        def g(*a):
            try:
                f(*a)
                return 0
            except Exception:   # noqa: BLE001 — must not unwind through C
                import traceback
                traceback.print_exc()
                return 1
        return g

    def _allgather(_u, h_send, nbytes, h_recv):
        parts = impl.allgather(C.string_at(h_send, nbytes))
        C.memmove(h_recv, b"".join(parts), nbytes * world)

    def _broadcast(_u, h_buf, nbytes, root):
        data = impl.broadcast(C.string_at(h_buf, nbytes) if rank == root else None, nbytes, root)
        C.memmove(h_buf, data, nbytes)

    def _alltoallv(_u, sp, soff, scnt, rp, roff, rcnt):
        impl.alltoallv(sp or 0, [soff[i] for i in range(world)], [scnt[i] for i in range(world)], rp or 0, [roff[i] for i in range(world)], [rcnt[i] for i in range(world)])

    def _abort():
        """nx_comm.abort: this rank's prove failed; make the peers' collectives fail instead of waiting (optional on the Python side)"""
        try:
            if hasattr(impl, "abort"):
                impl.abort()
        except Exception:   # noqa: BLE001 — must not unwind through C
            pass

    c = NxComm(rank, world, None, _SEND_T(guard(lambda _u, dst, p, n: impl.send(dst, p, n))), _RECV_T(guard(lambda _u, src, p, n: impl.recv(src, p, n))),
               _ALLREDUCE_T(guard(lambda _u, p, n: impl.allreduce_m31(p, n))), _ALLGATHER_T(guard(_allgather)), _BROADCAST_T(guard(_broadcast)),
               _ALLTOALLV_T(guard(_alltoallv)), _ALLGATHER_DEV_T(guard(lambda _u, sp, n, rp: impl.allgather_dev(sp or 0, n, rp or 0))),
               _ABORT_T(lambda _u: _abort()))
    c._impl = impl   # keep the callbacks' target alive
    return c


def rccl_unique_id():
    """ncclGetUniqueId through the library's native RCCL transport (rank 0 calls it and ships the 128 bytes to the other ranks)."""
    L = load_library()
    buf = (C.c_uint8 * 128)()
    rc = L.nx_rccl_unique_id(buf)
    if rc != 0:
        raise NexusHipError(f"nx_rccl_unique_id failed ({rc}): {L.nx_last_error(None).decode()}")
    return bytes(buf)


def _select_arg(select, n_c):
    if select is None:
        return None, None
    sel = np.ascontiguousarray(np.asarray(select, dtype=np.uint8).reshape(-1))
    assert len(sel) == n_c, "one flag per constraint"
    return sel, sel.ctypes.data_as(C.c_void_p)


def air_source(program, n_cols, select=None):
    """The HIP source nx_air_compile(_subset) generates for a recorded program (needs no GPU and no context).
    select: one flag per constraint (None = all)."""
    L = load_library()
    ins = _u32(program.instrs).reshape(-1)
    n_c = int(sum(1 for op in ins[0::4] if op in (13, 14)))
    src = C.c_void_p()
    sel, sel_p = _select_arg(select, n_c)
    rc = L.nx_air_compile_subset(None, ins.ctypes.data_as(C.c_void_p), len(ins) // 4, program.n_regs, n_cols, len(_u32(program.econsts).reshape(-1)) // 4, n_c, sel_p,
                                 None, C.byref(src))
    if rc != 0:
        raise NexusHipError(f"nx_air_compile failed ({rc}): {L.nx_last_error(None).decode()}")
    try:
        return C.string_at(src.value).decode()
    finally:
        L.nx_free_host(src)


def logup_program_source(program, n_cols, n_logup_cols):
    """The HIP source nx_logup_program generates for a fraction program (needs no GPU and no context)."""
    L = load_library()
    ins = _u32(program.instrs).reshape(-1)
    ec = _u32(program.econsts).reshape(-1)
    src = C.c_char_p()
    rc = L.nx_logup_program(None, ins.ctypes.data_as(C.c_void_p), len(ins) // 4, program.n_regs, None, n_cols, ec.ctypes.data_as(C.c_void_p) if len(ec) else None, len(ec) // 4,
                            4, n_logup_cols, None, C.byref(src))
    if rc != 0:
        raise NexusHipError(f"nx_logup_program failed ({rc}): {L.nx_last_error(None).decode()}")
    out = src.value.decode()
    L.nx_free_host(src)
    return out


def air_constraint_degrees(program, n_cols):
    """nx_air_constraint_degrees: an upper bound of every constraint's degree in the trace columns (host only)."""
    L = load_library()
    ins = _u32(program.instrs).reshape(-1)
    n_c = int(sum(1 for op in ins[0::4] if op in (13, 14)))
    out = np.zeros(max(1, n_c), np.uint32)
    rc = L.nx_air_constraint_degrees(None, ins.ctypes.data_as(C.c_void_p), len(ins) // 4, program.n_regs, n_cols, len(_u32(program.econsts).reshape(-1)) // 4, n_c,
                                     out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise NexusHipError(f"nx_air_constraint_degrees failed ({rc}): {L.nx_last_error(None).decode()}")
    return out[:n_c]


def air_cache_dir(path):
    """nx_air_cache_dir: the directory the library keeps compiled AIR kernels in (None = off).  Process-wide."""
    L = load_library()
    rc = L.nx_air_cache_dir(path.encode() if path else None)
    if rc != 0:
        raise NexusHipError(f"nx_air_cache_dir failed ({rc})")


def air_cache_stats():
    """(kernels compiled by hiprtc, loaded from the cache directory, stored into it) by this process"""
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    load_library().nx_air_cache_stats(C.byref(a), C.byref(b), C.byref(c))
    return int(a.value), int(b.value), int(c.value)


class AirKernel:
    """A recorded AIR compiled by hiprtc (nx_air_compile / nx_air_compile_subset); eval() matches HipBackend.eval_constraint_program.
    select: one flag per constraint — the kernel evaluates only those (columns they do not read may be passed as None to eval)."""

    def __init__(self, be, program, n_cols, select=None, blob=None):
        """blob: a kernel saved by save() (nx_air_kernel_load: no compilation) — the program is still needed for eval()'s default constants"""
        self.be, self.program, self.n_cols = be, program, n_cols
        ins = _u32(program.instrs).reshape(-1)
        self.n_constraints = int(sum(1 for op in ins[0::4] if op in (13, 14)))
        self.h = C.c_void_p()
        if blob is not None:
            b = (C.c_uint8 * len(blob)).from_buffer_copy(bytes(blob))
            be._chk(be.L.nx_air_kernel_load(be.ctx, b, C.c_size_t(len(blob)), C.byref(self.h)))
            return
        sel, sel_p = _select_arg(select, self.n_constraints)
        be._chk(be.L.nx_air_compile_subset(be.ctx, ins.ctypes.data_as(C.c_void_p), len(ins) // 4, program.n_regs, n_cols, len(_u32(program.econsts).reshape(-1)) // 4,
                                           self.n_constraints, sel_p, C.byref(self.h), None))

    def save(self):
        """The compiled kernel as bytes (nx_air_kernel_save): header + gfx950 code object, loadable by AirKernel(..., blob=...)."""
        p, n = C.c_void_p(), C.c_size_t()
        self.be._chk(self.be.L.nx_air_kernel_save(self.h, C.byref(p), C.byref(n)))
        out = C.string_at(p, n.value)
        self.be.L.nx_free_host(p)
        return out

    def eval(self, column_ptrs, alpha_powers, denom_inv, log_size, log_eval, acc4, econsts=None):
        assert len(column_ptrs) == self.n_cols
        ptrs = (C.c_void_p * max(1, len(column_ptrs)))(*column_ptrs)
        ec = _u32(self.program.econsts if econsts is None else econsts).reshape(-1)
        pw = _u32(alpha_powers).reshape(-1)
        assert len(pw) // 4 == self.n_constraints
        den = _u32(denom_inv)
        assert len(den) == 1 << (log_eval - log_size)
        self.be._chk(self.be.L.nx_air_eval(self.be.ctx, self.h, ptrs, ec.ctypes.data_as(C.c_void_p), pw.ctypes.data_as(C.c_void_p), den.ctypes.data_as(C.c_void_p),
                                           log_size, log_eval, acc4.col_ptrs()))
        return acc4

    def close(self):
        if self.h and self.be.ctx:
            self.be.L.nx_air_kernel_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AirComponentC(C.Structure):
    _fields_ = [("log_size", C.c_uint32), ("program", C.c_void_p), ("n_instr", C.c_uint32), ("n_regs", C.c_uint32), ("econsts", C.c_void_p),
                ("n_econsts", C.c_uint32), ("n_constraints", C.c_uint32), ("col_tree", C.c_void_p), ("col_index", C.c_void_p), ("n_cols", C.c_uint32),
                ("mask_count", C.c_void_p), ("mask_offsets", C.c_void_p), ("kernel", C.c_void_p), ("log_constraint_degree_bound", C.c_uint32)]


class ProverSession:
    """nx_prover_*: CommitmentSchemeProver + Blake2sChannel + stwo::prover::prove over recorded AIRs
    (air_program.Component).  The caller drives the reference's transcript prefix (machine.rs:198-263)."""

    def __init__(self, be, cfg, max_log_size):
        self.be, self.cfg = be, cfg
        self.h = C.c_void_p()
        be._chk(be.L.nx_prover_create(be.ctx, C.byref(cfg), max_log_size, C.byref(self.h)))

    def set_comm(self, comm):
        """ONE proof on several GPUs: every rank runs the same session calls; tree_begin then returns None for the columns of
        other ranks (nx_prover_set_comm).  Call before the first tree."""
        self._comm = comm
        self.be._chk(self.be.L.nx_prover_set_comm(self.h, C.byref(comm)))

    def mix_u64(self, v):
        self.be._chk(self.be.L.nx_prover_mix_u64(self.h, C.c_uint64(int(v))))

    def mix_felts(self, felts):
        f = _u32(felts).reshape(-1)
        self.be._chk(self.be.L.nx_prover_mix_felts(self.h, f.ctypes.data_as(C.c_void_p), len(f) // 4))

    def draw_felt(self):
        out = np.zeros(4, np.uint32)
        self.be._chk(self.be.L.nx_prover_draw_felt(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def draw_felts(self, n):
        out = np.zeros((n, 4), np.uint32)
        self.be._chk(self.be.L.nx_prover_draw_felts(self.h, n, out.ctypes.data_as(C.c_void_p)))
        return out

    def digest(self):
        out = np.zeros(8, np.uint32)
        self.be._chk(self.be.L.nx_prover_channel_digest(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def tree_begin(self, log_sizes):
        """Device addresses (ints) of the next tree's columns, to be filled with bit-reversed circle-domain evaluations."""
        logs = _u32(log_sizes)
        ptrs = (C.c_void_p * max(1, len(logs)))()
        self.be._chk(self.be.L.nx_prover_tree_begin(self.h, logs.ctypes.data_as(C.c_void_p), len(logs), ptrs))
        return [int(ptrs[i] or 0) for i in range(len(logs))]

    def tree_commit(self):
        root = np.zeros(8, np.uint32)
        self.be._chk(self.be.L.nx_prover_tree_commit(self.h, root.ctypes.data_as(C.c_void_p)))
        return root

    def commit(self, cols):
        """Host columns (numpy, bit-reversed circle-domain evaluations): tree_begin + upload + tree_commit; returns the root."""
        cols = [_u32(c) for c in cols]
        ptrs = self.tree_begin([int(np.log2(len(c))) for c in cols])
        for c, d in zip(cols, ptrs):
            if d:        # row-sharded prove: only this rank's columns are handed out
                self.be._chk(self.be.L.nx_upload(self.be.ctx, C.c_void_p(d), c.ctypes.data_as(C.c_void_p), C.c_size_t(len(c))))
        return self.tree_commit()

    def commit_host(self, cols, coset_order=False, keep=()):
        """nx_prover_tree_commit_host: the tree's columns stay in HOST memory (numpy) and are uploaded chunk by chunk under the commit's
        own transforms.  keep: column indices whose evaluations are cloned on arrival; returns (root, {index: DeviceColumns})."""
        cols = [_u32(c) for c in cols]
        logs = [int(np.log2(len(c))) for c in cols]
        self.tree_begin(logs)
        hp = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data for c in cols])
        kept = {int(k): DeviceColumns(self.be, 1, logs[int(k)]) for k in keep}
        ki = _u32(list(kept.keys()))
        kd = (C.c_void_p * max(1, len(kept)))(*[d.ptr.value for d in kept.values()])
        root = np.zeros(8, np.uint32)
        self.be._chk(self.be.L.nx_prover_tree_commit_host(self.h, hp, int(bool(coset_order)), ki.ctypes.data_as(C.c_void_p) if len(kept) else None, len(kept),
                                                         kd if len(kept) else None, root.ctypes.data_as(C.c_void_p)))
        return root, kept

    def share_tree(self, tree_index):
        """nx_prover_tree_share: committed tree `tree_index` of this session as a handle other sessions of the same backend can adopt."""
        h = C.c_void_p()
        self.be._chk(self.be.L.nx_prover_tree_share(self.h, int(tree_index), C.byref(h)))
        return SharedTree(self.be, h)

    def adopt_tree(self, shared):
        """nx_prover_tree_adopt: the shared tree becomes this session's next tree (no upload, no transform, no hashing); returns the root."""
        root = np.zeros(8, np.uint32)
        self.be._chk(self.be.L.nx_prover_tree_adopt(self.h, shared.h, root.ctypes.data_as(C.c_void_p)))
        return root

    def prove(self, components, kernels=None, want_stats=False):
        """components: air_program.Component list; kernels: optional AirKernel per component (compiled once, reused)."""
        arr = (AirComponentC * len(components))()
        keep = []
        for i, c in enumerate(components):
            pr = c.program
            ins, ec = _u32(pr.instrs).reshape(-1), _u32(pr.econsts).reshape(-1)
            ct, ci = _u32([t for t, _ in c.cols]), _u32([k for _, k in c.cols])
            mc = _u32([len(m) for m in c.masks])
            mo = np.ascontiguousarray([o for m in c.masks for o in m], dtype=np.int32)
            keep += [ins, ec, ct, ci, mc, mo]
            a = arr[i]
            a.log_size, a.program, a.n_instr, a.n_regs = c.log_size, ins.ctypes.data, len(ins) // 4, pr.n_regs
            a.econsts, a.n_econsts, a.n_constraints = ec.ctypes.data if len(ec) else None, len(ec) // 4, pr.n_constraints
            a.col_tree, a.col_index, a.n_cols = ct.ctypes.data if len(ct) else None, ci.ctypes.data if len(ci) else None, len(c.cols)
            a.mask_count, a.mask_offsets = mc.ctypes.data if len(mc) else None, mo.ctypes.data if len(mo) else None
            a.kernel = kernels[i].h if kernels else None
            a.log_constraint_degree_bound = getattr(c, "log_constraint_degree_bound", 0)
        words, n = C.POINTER(C.c_uint32)(), C.c_size_t(0)
        stats = ProveStats()
        self.be._chk(self.be.L.nx_prover_prove(self.h, arr, len(components), C.byref(words), C.byref(n), C.byref(stats) if want_stats else None))
        out = np.ctypeslib.as_array(words, shape=(n.value,)).copy()
        self.be.L.nx_free_host(words)
        return (out, stats.as_dict()) if want_stats else out

    def close(self):
        if self.h and self.be.ctx:
            self.be.L.nx_prover_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SharedTree:
    """A committed tree shared between sessions (nx_committed_tree): release() when no further session will adopt it."""

    def __init__(self, be, h):
        self.be, self.h = be, h

    def root(self):
        r, n = np.zeros(8, np.uint32), C.c_uint32()
        self.be._chk(self.be.L.nx_committed_tree_root(self.h, r.ctypes.data_as(C.c_void_p), C.byref(n)))
        return r, int(n.value)

    def release(self):
        if self.h and self.be.ctx:
            self.be.L.nx_committed_tree_release(self.h)
        self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class LogupFrac(C.Structure):
    """nx_logup_frac of include/nexus_hip.h."""
    _fields_ = [("d_tuple_cols", C.c_void_p), ("n_tuple_cols", C.c_uint32), ("alpha_powers", C.c_void_p), ("z", C.c_void_p), ("d_mult", C.c_void_p),
                ("scale", C.c_void_p)]


class ProveStats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("trace_gen", "commit", "composition", "oods", "quotients", "fri", "pow", "decommit", "total")] + \
               [("lde_kernel_ms", C.c_double), ("lde_algorithmic_bytes", C.c_uint64), ("merkle_kernel_ms", C.c_double),
                ("merkle_algorithmic_bytes", C.c_uint64), ("interaction", C.c_double), ("comm_ms", C.c_double), ("comm_bytes", C.c_uint64),
                ("n_alltoallv", C.c_uint32), ("n_allgather_dev", C.c_uint32), ("n_allgather_host", C.c_uint32), ("n_comm_reserved", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def load_library():
    """Loads libnexus_hip.so (built in-tree by __graft_entry__.build / csrc/Makefile).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NexusHipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.nx_last_error.restype = C.c_char_p
    L.nx_last_error.argtypes = [C.c_void_p]
    L.nx_version.restype = C.c_char_p
    L.nx_ctx_stream.restype = C.c_void_p
    L.nx_merkle_layer.restype = C.c_void_p
    L.nx_merkle_n_layers.restype = C.c_uint32
    L.nx_free_host.argtypes = [C.c_void_p]
    L.nx_air_kernel_destroy.argtypes = [C.c_void_p]
    L.nx_prover_destroy.argtypes = [C.c_void_p]
    L.nx_comm_group_destroy.argtypes = [C.c_void_p]
    L.nx_comm_local_destroy.argtypes = [C.c_void_p]
    L.nx_committed_tree_release.argtypes = [C.c_void_p]
    L.nx_committed_tree_release.restype = None
    for name in ("nx_ctx_destroy", "nx_twiddles_destroy", "nx_tree_destroy", "nx_free_host", "nx_air_kernel_destroy", "nx_prover_destroy", "nx_comm_group_destroy",
                 "nx_comm_local_destroy", "nx_comm_rccl_destroy"):
        getattr(L, name).restype = None
    _lib = L
    return L


def declared_symbols():
    """Function names declared in include/nexus_hip.h (used by the CPU test that the .so exports them all)."""
    import re
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nx_[a-z0-9_]+)\s*\(", text)))


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def default_config(pow_bits=10, log_blowup=1, n_queries=3, log_last_layer_degree_bound=0, hash_mode=HASH_BLAKE2S,
                   fri_alpha_mode=FRI_ALPHA_PREV, log_constraint_degree=1):
    """PcsConfig::default() of the pinned Stwo is unverifiable here (pow_bits 5 or 10)."""
    return PcsConfig(pow_bits, log_blowup, n_queries, log_last_layer_degree_bound, hash_mode, fri_alpha_mode, log_constraint_degree)


class DeviceColumns:
    """n_cols contiguous device columns of 2^log_size u32 words (a slab)."""

    def __init__(self, backend, n_cols, log_size):
        self.be, self.n_cols, self.log_size = backend, n_cols, log_size
        self.ptr = C.c_void_p()
        backend._chk(backend.L.nx_alloc(backend.ctx, C.c_size_t(max(1, n_cols << log_size)), C.byref(self.ptr)))

    @classmethod
    def view(cls, backend, ptr, n_cols, log_size):
        """A non-owning view of n_cols contiguous columns that live elsewhere (e.g. in a ProverSession tree)."""
        v = cls.__new__(cls)
        v.be, v.n_cols, v.log_size, v.ptr, v.borrowed = backend, n_cols, log_size, C.c_void_p(int(ptr)), True
        return v

    def col_ptrs(self):
        stride = 4 << self.log_size
        return (C.c_void_p * max(1, self.n_cols))(*[self.ptr.value + i * stride for i in range(self.n_cols)])

    def upload(self, arr2d):
        a = _u32(arr2d).reshape(self.n_cols, 1 << self.log_size)
        self.be._chk(self.be.L.nx_upload(self.be.ctx, self.ptr, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size)))
        return self

    def to_cpu(self):
        out = np.empty((self.n_cols, 1 << self.log_size), np.uint32)
        if out.size:
            self.be._chk(self.be.L.nx_download(self.be.ctx, out.ctypes.data_as(C.c_void_p), self.ptr, C.c_size_t(out.size)))
        return out

    def free(self):
        if getattr(self, "borrowed", False):
            self.ptr = C.c_void_p()
            return
        if self.ptr and self.ptr.value:
            if self.be.ctx:   # after HipBackend.close() the context (and every allocation it cached) is gone
                self.be.L.nx_free(self.be.ctx, self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Twiddles:
    def __init__(self, backend, log_half_coset):
        self.be, self.log_half = backend, log_half_coset
        self.h = C.c_void_p()
        backend._chk(backend.L.nx_twiddles_create(backend.ctx, log_half_coset, C.byref(self.h)))

    def to_cpu(self):
        n = 1 << self.log_half
        tw, itw = np.empty(n, np.uint32), np.empty(n, np.uint32)
        self.be._chk(self.be.L.nx_twiddles_download(self.be.ctx, self.h, tw.ctypes.data_as(C.c_void_p), itw.ctypes.data_as(C.c_void_p)))
        return tw, itw

    def __del__(self):
        try:
            if self.h:
                if self.be.ctx:
                    self.be.L.nx_twiddles_destroy(self.h)
                self.h = None
        except Exception:
            pass


class MerkleTree:
    def __init__(self, backend, handle):
        self.be, self.h = backend, handle

    def root(self):
        out = np.empty(8, np.uint32)
        self.be._chk(self.be.L.nx_merkle_root(self.be.ctx, self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def layer(self, k):
        n = 8 << k
        out = np.empty(n, np.uint32)
        p = C.c_void_p(self.be.L.nx_merkle_layer(self.h, k))
        self.be._chk(self.be.L.nx_download(self.be.ctx, out.ctypes.data_as(C.c_void_p), p, C.c_size_t(n)))
        return out.reshape(-1, 8)

    def n_layers(self):
        return self.be.L.nx_merkle_n_layers(self.h)

    def __del__(self):
        try:
            if self.h:
                if self.be.ctx:
                    self.be.L.nx_tree_destroy(self.h)
                self.h = None
        except Exception:
            pass


class HipBackend:
    """One context = one GPU = one prover transcript (the reference's single &mut channel)."""

    def __init__(self, device=0):
        self.L = load_library()
        self.device = int(device)
        self.ctx = C.c_void_p()
        rc = self.L.nx_ctx_create(int(device), C.byref(self.ctx))
        if rc != NX_OK:
            raise NexusHipError(f"nx_ctx_create failed ({rc}): {self.L.nx_last_error(None).decode()}")

    def _chk(self, rc):
        if rc != NX_OK:
            raise NexusHipError(f"libnexus_hip error {rc}: {self.L.nx_last_error(self.ctx).decode()}")

    def close(self):
        if self.ctx:
            self.L.nx_ctx_destroy(self.ctx)
            self.ctx = None

    def sync(self):
        self._chk(self.L.nx_sync(self.ctx))

    def trim(self):
        """nx_ctx_trim: give the cached device blocks back to the driver (other contexts on this GPU can use them)."""
        self._chk(self.L.nx_ctx_trim(self.ctx))

    def set_hash_mode(self, mode):
        self._chk(self.L.nx_ctx_set_hash_mode(self.ctx, mode))

    def rccl_comm(self, unique_id, rank, world):
        """The native RCCL transport (csrc/comm_rccl.hip) as an NxComm for prove_machine(comm=...) / ProverSession.set_comm: no Python in
        the data path.  Free with free_rccl_comm."""
        p = C.c_void_p()
        idb = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._chk(self.L.nx_comm_rccl_create(self.ctx, idb, int(rank), int(world), C.byref(p)))
        comm = NxComm.from_address(p.value)
        comm._native = p
        return comm

    def free_rccl_comm(self, comm):
        self.L.nx_comm_rccl_destroy(comm._native)

    def local_comm(self, group, rank):
        """The in-process transport (csrc/comm_local.hip) as an NxComm: `group` = LocalGroup(world) shared by the threads of one proof,
        one thread per rank with a HipBackend of its own.  No Python in the data path.  Free with free_local_comm."""
        p = C.c_void_p()
        self._chk(self.L.nx_comm_local_create(group.h, self.ctx, int(rank), C.byref(p)))
        comm = NxComm.from_address(p.value)
        comm._native, comm._group = p, group
        return comm

    def free_local_comm(self, comm):
        self.L.nx_comm_local_destroy(comm._native)

    def set_option(self, name, value):
        """Per-context policy / tuning (nx_ctx_set_option): "fft.batch_cols", "fft.streams",
        "fri.dist_min_log", "dist.chunks", "air.segment", ...  The NX_* environment variables only seed a new context's defaults."""
        self._chk(self.L.nx_ctx_set_option(self.ctx, name.encode(), C.c_int64(int(value))))

    def get_option(self, name):
        v = C.c_int64()
        self._chk(self.L.nx_ctx_get_option(self.ctx, name.encode(), C.byref(v)))
        return v.value

    # ---- Column ops ----
    def clone_columns(self, cols):
        """Column::clone on device (nx_copy)."""
        out = DeviceColumns(self, cols.n_cols, cols.log_size)
        self._chk(self.L.nx_copy(self.ctx, out.ptr, cols.ptr, C.c_size_t(cols.n_cols << cols.log_size)))
        return out

    def columns(self, n_cols, log_size):
        return DeviceColumns(self, n_cols, log_size)

    def columns_from_host(self, arr2d):
        a = _u32(arr2d)
        if a.ndim == 1:
            a = a[None, :]
        return DeviceColumns(self, a.shape[0], int(np.log2(a.shape[1]))).upload(a)

    def bit_reverse_column(self, cols, index=0):
        p = C.c_void_p(cols.ptr.value + index * (4 << cols.log_size))
        self._chk(self.L.nx_bit_reverse(self.ctx, p, cols.log_size))

    def finalize_columns(self, natural):
        """R3: natural coset order -> bit-reversed circle-domain order (reference trace/utils.rs:94-106)."""
        out = DeviceColumns(self, natural.n_cols, natural.log_size)
        self._chk(self.L.nx_finalize_columns(self.ctx, natural.col_ptrs(), out.col_ptrs(), natural.n_cols, natural.log_size))
        return out

    def upload_coset_order(self, host_col):
        a = _u32(host_col)
        out = DeviceColumns(self, 1, int(np.log2(a.size)))
        self._chk(self.L.nx_upload_coset_order(self.ctx, a.ctypes.data_as(C.c_void_p), out.log_size, out.ptr))
        return out

    def upload_columns(self, host_cols, coset_order=True):
        """A whole host trace (list of 1-D uint32 arrays of one size) -> DeviceColumns, pinned in place and streamed
        (nx_upload_columns); coset_order: the host holds natural coset order (reference trace builders) and wants the
        bit-reversed circle-domain order on device."""
        cols = [_u32(c) for c in host_cols]
        log = int(np.log2(cols[0].size))
        out = DeviceColumns(self, len(cols), log)
        hp = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
        self._chk(self.L.nx_upload_columns(self.ctx, hp, len(cols), log, out.col_ptrs(), 1 if coset_order else 0))
        return out

    def host_pin(self, arr):
        """nx_host_pin: registers a contiguous host array with the driver once (a trace buffer reused across proofs); the entry points
        that take host columns then skip their own per-call pinning of it.  Undo with host_unpin before the array is freed."""
        a = np.ascontiguousarray(arr)
        if a is not arr and not np.shares_memory(a, arr):
            raise ValueError("host_pin needs a contiguous array (it pins the memory in place)")
        self._chk(self.L.nx_host_pin(self.ctx, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes)))

    def host_unpin(self, arr):
        self._chk(self.L.nx_host_unpin(self.ctx, C.c_void_p(np.ascontiguousarray(arr).ctypes.data)))

    # ---- PolyOps ----
    def precompute_twiddles(self, log_half_coset):
        return Twiddles(self, log_half_coset)

    def interpolate_columns(self, tw, cols):
        self._chk(self.L.nx_interpolate_batch(self.ctx, tw.h, cols.col_ptrs(), cols.n_cols, cols.log_size))
        return cols

    def evaluate_polynomials(self, tw, polys, log_expand):
        out = DeviceColumns(self, polys.n_cols, polys.log_size + log_expand)
        self._chk(self.L.nx_evaluate_batch(self.ctx, tw.h, polys.col_ptrs(), polys.n_cols, polys.log_size, log_expand, out.col_ptrs()))
        return out

    def lde(self, tw, cols, log_blowup):
        out = DeviceColumns(self, cols.n_cols, cols.log_size + log_blowup)
        self._chk(self.L.nx_lde_batch(self.ctx, tw.h, cols.col_ptrs(), cols.n_cols, cols.log_size, log_blowup, out.col_ptrs()))
        return out

    def lde_commit(self, tw, cols, log_blowup):
        out = DeviceColumns(self, cols.n_cols, cols.log_size + log_blowup)
        root = np.empty(8, np.uint32)
        self._chk(self.L.nx_lde_commit(self.ctx, tw.h, cols.col_ptrs(), cols.n_cols, cols.log_size, log_blowup, out.col_ptrs(),
                                       root.ctypes.data_as(C.c_void_p)))
        return out, root

    def eval_at_points(self, polys, poly_idx, points):
        idx = _u32(poly_idx)
        pts = _u32(points).reshape(-1, 8)
        out = np.empty((len(idx), 4), np.uint32)
        self._chk(self.L.nx_eval_at_points(self.ctx, polys.col_ptrs(), polys.log_size, idx.ctypes.data_as(C.c_void_p),
                                           pts.ctypes.data_as(C.c_void_p), len(idx), out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- MerkleOps ----
    def merkle_commit(self, column_sets):
        """column_sets: list of DeviceColumns in commit order (mixed sizes allowed)."""
        ptrs, logs = [], []
        for cs in column_sets:
            stride = 4 << cs.log_size
            for i in range(cs.n_cols):
                ptrs.append(cs.ptr.value + i * stride)
                logs.append(cs.log_size)
        arr = (C.c_void_p * max(1, len(ptrs)))(*ptrs)
        lg = _u32(logs)
        h = C.c_void_p()
        self._chk(self.L.nx_merkle_commit(self.ctx, arr, lg.ctypes.data_as(C.c_void_p), len(ptrs), C.byref(h)))
        return MerkleTree(self, h)

    def merkle_leaf_chain(self, cols, col_offset, total_cols, state_in_ptr, state_out_ptr, row_begin=0, n_rows=None):
        """One column shard of a leaf layer (nx_merkle_leaf_chain): cols are this shard's columns (a DeviceColumns),
        state pointers are raw device addresses (ints) of n_rows x 8 words; state_in_ptr is None for the shard at column 0."""
        if n_rows is None:
            n_rows = (1 << cols.log_size) - row_begin
        self._chk(self.L.nx_merkle_leaf_chain(self.ctx, cols.col_ptrs(), cols.n_cols, cols.log_size, col_offset, total_cols,
                                              C.c_void_p(state_in_ptr) if state_in_ptr else None, C.c_void_p(state_out_ptr),
                                              C.c_uint64(row_begin), C.c_uint64(n_rows)))

    def merkle_from_leaves(self, leaf_ptr, log_size):
        """Inner layers above 2^log_size leaf digests at device address leaf_ptr (nx_merkle_from_leaves)."""
        h = C.c_void_p()
        self._chk(self.L.nx_merkle_from_leaves(self.ctx, C.c_void_p(leaf_ptr), log_size, C.byref(h)))
        return MerkleTree(self, h)


    # ---- the remaining Backend supertraits: AccumulationOps, FieldOps::batch_inverse, per-layer MerkleOps,
    #      MerkleProver::decommit, FriOps::decompose ----
    def secure_accumulate(self, dst4, src4):
        self._chk(self.L.nx_secure_accumulate(self.ctx, dst4.col_ptrs(), src4.col_ptrs(), dst4.log_size))
        return dst4

    def generate_secure_powers(self, felt, n):
        out = np.zeros((n, 4), np.uint32)
        f = _u32(felt)
        self._chk(self.L.nx_generate_secure_powers(f.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p)))
        return out

    def batch_inverse_m31(self, cols):
        """Element-wise inverse of every word of a DeviceColumns (FieldOps<BaseField>::batch_inverse)."""
        out = DeviceColumns(self, cols.n_cols, cols.log_size)
        self._chk(self.L.nx_batch_inverse_m31(self.ctx, cols.ptr, out.ptr, C.c_size_t(cols.n_cols << cols.log_size)))
        return out

    def batch_inverse_qm31(self, cols4):
        out = DeviceColumns(self, 4, cols4.log_size)
        self._chk(self.L.nx_batch_inverse_qm31(self.ctx, cols4.col_ptrs(), out.col_ptrs(), C.c_size_t(1 << cols4.log_size)))
        return out

    def bit_reverse_secure(self, cols4):
        self._chk(self.L.nx_bit_reverse_secure(self.ctx, cols4.col_ptrs(), cols4.log_size))
        return cols4

    def merkle_commit_on_layer(self, log_size, prev_layer_ptr, cols):
        """One Merkle layer (MerkleOps::commit_on_layer): prev_layer_ptr = device address of the 2^(log_size+1) nodes below or None;
        cols = DeviceColumns of 2^log_size rows injected here (or None).  Returns a DeviceColumns view-like buffer of 8 * 2^log_size words."""
        out = DeviceColumns(self, 8, log_size)          # 2^log nodes x 8 words, node-major
        n = cols.n_cols if cols is not None else 0
        self._chk(self.L.nx_merkle_commit_on_layer(self.ctx, log_size, C.c_void_p(prev_layer_ptr) if prev_layer_ptr else None,
                                                   cols.col_ptrs() if n else None, n, out.ptr))
        return out

    def merkle_decommit(self, tree, column_sets, queries_per_log):
        """MerkleProver::decommit: queries_per_log = {log: sorted positions}.  Returns (queried_values, hash_witness[n, 8], column_witness)."""
        ptrs, logs = [], []
        for cs in column_sets:
            stride = 4 << cs.log_size
            for i in range(cs.n_cols):
                ptrs.append(cs.ptr.value + i * stride)
                logs.append(cs.log_size)
        arr = (C.c_void_p * max(1, len(ptrs)))(*ptrs)
        lg = _u32(logs)
        qlogs = _u32(sorted(queries_per_log))
        qcnt = _u32([len(queries_per_log[int(l)]) for l in qlogs])
        qs = np.ascontiguousarray([q for l in qlogs for q in queries_per_log[int(l)]], dtype=np.uint64)
        outs = [C.POINTER(C.c_uint32)() for _ in range(3)]
        ns = [C.c_size_t(0) for _ in range(3)]
        self._chk(self.L.nx_merkle_decommit(self.ctx, tree.h, arr, lg.ctypes.data_as(C.c_void_p), len(ptrs), qlogs.ctypes.data_as(C.c_void_p),
                                            qcnt.ctypes.data_as(C.c_void_p), len(qlogs), qs.ctypes.data_as(C.c_void_p),
                                            C.byref(outs[0]), C.byref(ns[0]), C.byref(outs[1]), C.byref(ns[1]), C.byref(outs[2]), C.byref(ns[2])))
        res = []
        for o, n, mul in zip(outs, ns, (1, 8, 1)):
            a = np.ctypeslib.as_array(o, shape=(max(1, n.value * mul),))[:n.value * mul].copy()
            self.L.nx_free_host(o)
            res.append(a)
        return res[0], res[1].reshape(-1, 8), res[2]

    def fri_decompose(self, src4):
        g = DeviceColumns(self, 4, src4.log_size)
        lam = np.zeros(4, np.uint32)
        self._chk(self.L.nx_fri_decompose(self.ctx, src4.col_ptrs(), src4.log_size, g.col_ptrs(), lam.ctypes.data_as(C.c_void_p)))
        return g, lam

    # ---- QuotientOps ----
    def accumulate_quotients(self, cols, random_coeff, batches):
        """batches: list of (point8, [(col_idx, value4), ...]).  Returns a 4-column DeviceColumns."""
        out = DeviceColumns(self, 4, cols.log_size)
        pts = _u32([b[0] for b in batches]).reshape(-1)
        counts = _u32([len(b[1]) for b in batches])
        cidx = _u32([cv[0] for b in batches for cv in b[1]])
        vals = _u32([cv[1] for b in batches for cv in b[1]]).reshape(-1)
        a = _u32(random_coeff)
        self._chk(self.L.nx_accumulate_quotients(self.ctx, cols.log_size, cols.col_ptrs(), cols.n_cols, a.ctypes.data_as(C.c_void_p),
                                                 len(batches), pts.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p),
                                                 cidx.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), out.col_ptrs()))
        return out

    # ---- recorded AIR constraints ----
    def eval_constraint_program(self, program, column_ptrs, alpha_powers, denom_inv, log_size, log_eval, acc4):
        """Run an air_program.Program over every row of the evaluation domain and accumulate into acc4 (a 4-column
        DeviceColumns of log size log_eval).  column_ptrs: device addresses (ints) of the columns the program's LOADs index."""
        ins = _u32(program.instrs).reshape(-1)
        ptrs = (C.c_void_p * max(1, len(column_ptrs)))(*column_ptrs)
        ec = _u32(program.econsts).reshape(-1)
        pw = _u32(alpha_powers).reshape(-1)
        den = _u32(denom_inv)
        self._chk(self.L.nx_eval_constraint_program(self.ctx, ins.ctypes.data_as(C.c_void_p), len(ins) // 4, program.n_regs, ptrs, len(column_ptrs),
                                                    ec.ctypes.data_as(C.c_void_p), len(ec) // 4, pw.ctypes.data_as(C.c_void_p), len(pw) // 4,
                                                    den.ctypes.data_as(C.c_void_p), log_size, log_eval, acc4.col_ptrs()))
        return acc4

    def prover_session(self, cfg, max_log_size):
        return ProverSession(self, cfg, max_log_size)

    def compile_air(self, program, n_cols, select=None):
        """nx_air_compile(_subset): the recorded program as a run-time-compiled gfx950 kernel (same semantics as the interpreter)."""
        return AirKernel(self, program, n_cols, select)

    # ---- logup interaction trace ----
    @staticmethod
    def _ptr4(cols4):
        return cols4.col_ptrs() if cols4 is not None else None

    def logup_combine(self, tuple_cols, alpha_powers, z):
        """Relation::combine over device columns: sum_i alpha_powers[i] * tuple_i - z -> a secure (4-column) DeviceColumns."""
        out = DeviceColumns(self, 4, tuple_cols.log_size)
        ap, zz = _u32(alpha_powers).reshape(-1), _u32(z)
        self._chk(self.L.nx_logup_combine(self.ctx, tuple_cols.col_ptrs(), tuple_cols.n_cols, ap.ctypes.data_as(C.c_void_p), zz.ctypes.data_as(C.c_void_p),
                                          tuple_cols.log_size, out.col_ptrs()))
        return out

    def logup_finalize_col(self, den_a, scale_a=(1, 0, 0, 0), mult_a=None, den_b=None, scale_b=(1, 0, 0, 0), mult_b=None, prev=None, out=None):
        """LogupColGenerator::write_frac + finalize_col (one fraction, or two merged as prover2 does).  mult_*: a 1-column
        DeviceColumns (numerator = scale * mult) or None (numerator = scale)."""
        out = out or DeviceColumns(self, 4, den_a.log_size)
        sa, sb = _u32(scale_a), _u32(scale_b)
        self._chk(self.L.nx_logup_finalize_col(self.ctx, den_a.log_size, mult_a.ptr if mult_a is not None else None, sa.ctypes.data_as(C.c_void_p),
                                               den_a.col_ptrs(), mult_b.ptr if mult_b is not None else None, sb.ctypes.data_as(C.c_void_p),
                                               self._ptr4(den_b), self._ptr4(prev), out.col_ptrs()))
        return out

    def logup_col(self, frac_a, frac_b=None, prev=None, out=None):
        """Fused combine + write_frac + finalize_col (nx_logup_col).  A fraction is a dict: tuple (DeviceColumns), alphas
        (n x 4), z (4), optional mult (1-column DeviceColumns), optional scale (4, default 1)."""
        keep = []

        def mk(f):
            t = f["tuple"]
            ptrs = t.col_ptrs(); ap = _u32(f["alphas"]).reshape(-1); z = _u32(f["z"]); sc = _u32(f.get("scale", (1, 0, 0, 0)))
            keep.extend([ptrs, ap, z, sc])
            m = f.get("mult")
            return LogupFrac(C.cast(ptrs, C.c_void_p), t.n_cols, ap.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p),
                             m.ptr if m is not None else None, sc.ctypes.data_as(C.c_void_p))
        fa = mk(frac_a)
        fb = mk(frac_b) if frac_b is not None else None
        out = out or DeviceColumns(self, 4, frac_a["tuple"].log_size)
        self._chk(self.L.nx_logup_col(self.ctx, frac_a["tuple"].log_size, C.byref(fa), C.byref(fb) if fb is not None else None, self._ptr4(prev), out.col_ptrs()))
        return out

    def logup_cols(self, fracs):
        """All the logup columns of a component in one launch (nx_logup_cols): column j = sum of the first j + 1 fractions of the
        row.  fracs: dicts as for logup_col.  Returns one 4-column DeviceColumns per fraction."""
        keep, arr = [], (LogupFrac * len(fracs))()
        for i, f in enumerate(fracs):
            t = f["tuple"]
            ptrs = t.col_ptrs(); ap = _u32(f["alphas"]).reshape(-1); z = _u32(f["z"]); sc = _u32(f.get("scale", (1, 0, 0, 0)))
            keep.extend([ptrs, ap, z, sc])
            m = f.get("mult")
            arr[i] = LogupFrac(C.cast(ptrs, C.c_void_p), t.n_cols, ap.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p),
                               m.ptr if m is not None else None, sc.ctypes.data_as(C.c_void_p))
        log = fracs[0]["tuple"].log_size
        outs = [DeviceColumns(self, 4, log) for _ in fracs]
        ptrs = (C.c_void_p * (4 * len(fracs)))(*[o.ptr.value + k * (4 << log) for o in outs for k in range(4)])
        self._chk(self.L.nx_logup_cols(self.ctx, log, arr, len(fracs), ptrs))
        return outs

    def logup_cols_batched(self, fracs, batching=None, n_cols=None):
        """finalize_logup_batched on the trace side (nx_logup_cols_batched): fraction i belongs to batch batching[i] (None: in pairs,
        i // 2), column j = the sum of the fractions of batches <= j.  Returns one 4-column DeviceColumns per logup column."""
        keep, arr = [], (LogupFrac * len(fracs))()
        for i, f in enumerate(fracs):
            t = f["tuple"]
            ptrs = t.col_ptrs(); ap = _u32(f["alphas"]).reshape(-1); z = _u32(f["z"]); sc = _u32(f.get("scale", (1, 0, 0, 0)))
            keep.extend([ptrs, ap, z, sc])
            m = f.get("mult")
            arr[i] = LogupFrac(C.cast(ptrs, C.c_void_p), t.n_cols, ap.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p),
                               m.ptr if m is not None else None, sc.ctypes.data_as(C.c_void_p))
        log = fracs[0]["tuple"].log_size
        if n_cols is None:
            n_cols = (len(fracs) + 1) // 2 if batching is None else max(batching) + 1
        outs = [DeviceColumns(self, 4, log) for _ in range(n_cols)]
        ptrs = (C.c_void_p * (4 * n_cols))(*[o.ptr.value + k * (4 << log) for o in outs for k in range(4)])
        b = None if batching is None else _u32(batching)
        self._chk(self.L.nx_logup_cols_batched(self.ctx, log, arr, len(fracs), b.ctypes.data_as(C.c_void_p) if b is not None else None, n_cols, ptrs))
        return outs

    def logup_program(self, program, column_ptrs, log_size, n_logup_cols=None, econsts=None, out_ptrs=None):
        """The interaction trace of a component from the relation entries its recorded AIR declares (nx_logup_program): program =
        ProgramBuilder.build_logup(); column_ptrs: one device pointer per component column (None where the program loads nothing).
        Returns one 4-column DeviceColumns per logup column (follow with logup_finalize_last on the last one) — or, with out_ptrs
        (4 device pointers per logup column, e.g. a session's own interaction-tree columns from tree_begin), writes there and returns
        out_ptrs: nothing is copied afterwards (rust/nexus-hip/src/simd_host.rs interaction_tree_on_device)."""
        n_logup_cols = program.n_logup_cols if n_logup_cols is None else n_logup_cols
        ins = _u32(program.instrs).reshape(-1)
        ec = _u32(program.econsts if econsts is None else econsts).reshape(-1)
        ptrs = (C.c_void_p * max(1, len(column_ptrs)))(*column_ptrs)
        if out_ptrs is None:
            outs = [DeviceColumns(self, 4, log_size) for _ in range(n_logup_cols)]
            flat = [o.ptr.value + k * (4 << log_size) for o in outs for k in range(4)]
        else:
            outs, flat = out_ptrs, [int(x) for x in out_ptrs]
            if len(flat) != 4 * n_logup_cols:
                raise NexusHipError(f"logup_program: {len(flat)} output columns for {n_logup_cols} logup columns")
        optr = (C.c_void_p * max(1, 4 * n_logup_cols))(*flat)
        self._chk(self.L.nx_logup_program(self.ctx, ins.ctypes.data_as(C.c_void_p), len(ins) // 4, program.n_regs, ptrs, len(column_ptrs),
                                          ec.ctypes.data_as(C.c_void_p) if len(ec) else None, len(ec) // 4, log_size, n_logup_cols, optr, None))
        return outs

    def logup_finalize_last(self, col4, log_size=None):
        """LogupTraceGenerator::finalize_last in place; returns the claimed sum (4 words).  col4: a 4-column DeviceColumns, or (with
        log_size) the 4 device pointers of the coordinate columns."""
        cs = np.zeros(4, np.uint32)
        if log_size is None:
            log_size, ptrs = col4.log_size, col4.col_ptrs()
        else:
            ptrs = (C.c_void_p * 4)(*[int(x) for x in col4])
        self._chk(self.L.nx_logup_finalize_last(self.ctx, log_size, ptrs, cs.ctypes.data_as(C.c_void_p)))
        return cs

    def logup_finalize_last_batch(self, cols4_list):
        """finalize_last for several secure columns of one size in one call; returns the claimed sums (n x 4)."""
        n = len(cols4_list)
        ptrs = (C.c_void_p * max(1, 4 * n))(*[c.ptr.value + q * (4 << c.log_size) for c in cols4_list for q in range(4)])
        cs = np.zeros((n, 4), np.uint32)
        self._chk(self.L.nx_logup_finalize_last_batch(self.ctx, cols4_list[0].log_size if n else 0, ptrs, n, cs.ctypes.data_as(C.c_void_p)))
        return cs

    # ---- FriOps ----
    def fold_circle_into_line(self, tw, dst4, src4, alpha):
        a = _u32(alpha)
        self._chk(self.L.nx_fold_circle_into_line(self.ctx, tw.h, dst4.col_ptrs(), src4.col_ptrs(), src4.log_size, a.ctypes.data_as(C.c_void_p)))
        return dst4

    def fold_line(self, tw, src4, alpha, n_doublings=0):
        out = DeviceColumns(self, 4, src4.log_size - 1)
        a = _u32(alpha)
        self._chk(self.L.nx_fold_line(self.ctx, tw.h, src4.col_ptrs(), src4.log_size, n_doublings, a.ctypes.data_as(C.c_void_p), out.col_ptrs()))
        return out

    # ---- GrindOps ----
    def grind(self, digest_words, pow_bits):
        d = _u32(digest_words)
        nonce = C.c_uint64()
        self._chk(self.L.nx_grind(self.ctx, d.ctypes.data_as(C.c_void_p), pow_bits, C.byref(nonce)))
        return nonce.value

    # ---- synthetic machine ----
    @staticmethod
    def _comps(comps):
        return (ComponentSpec * len(comps))(*[ComponentSpec(*[int(x) for x in c]) for c in comps])   # 4-, 5- or 6-tuples (bound and logup_mode default to 0)

    def synth_fill_tree(self, comps, tree, seed, inter_seed=0):
        sets = []
        ptrs = []
        for (ls, a, b, c) in [tuple(x)[:4] for x in comps]:
            n = [a, b, c][tree]
            s = DeviceColumns(self, n, ls)
            sets.append(s)
            ptrs += [s.ptr.value + i * (4 << ls) for i in range(n)]
        arr = (C.c_void_p * max(1, len(ptrs)))(*ptrs)
        self._chk(self.L.nx_synth_fill_tree(self.ctx, self._comps(comps), len(comps), tree, C.c_uint64(seed), C.c_uint64(inter_seed), arr))
        return sets

    def prove(self, comps, cfg=None, seed=1, ad=b"", want_stats=False):
        """nexus_vm_prover::prove analogue for the synthetic machine (reference prover/src/lib.rs:26-31)."""
        cfg = cfg or default_config()
        words, n = C.POINTER(C.c_uint32)(), C.c_size_t(0)
        stats = ProveStats()
        adb = (C.c_uint8 * max(1, len(ad)))(*ad)
        self._chk(self.L.nx_prove_synth(self.ctx, self._comps(comps), len(comps), C.byref(cfg), C.c_uint64(seed), adb, C.c_size_t(len(ad)),
                                        C.byref(words), C.byref(n), C.byref(stats) if want_stats else None))
        out = np.ctypeslib.as_array(words, shape=(n.value,)).copy()
        self.L.nx_free_host(words)
        return (out, stats.as_dict()) if want_stats else out

    def prove_machine(self, comps, cfg=None, seed=1, ad=b"", want_stats=False, comm=None):
        """nexus_vm_prover::prove analogue with a REAL logup interaction trace and a recorded AIR (nx_prove_machine); comps:
        (log_size, n_pre, n_main, 4 x logup columns).  comm: an NxComm for ONE proof on several GPUs (every rank calls this)."""
        cfg = cfg or default_config()
        words, n = C.POINTER(C.c_uint32)(), C.c_size_t(0)
        stats = ProveStats()
        adb = (C.c_uint8 * max(1, len(ad)))(*ad)
        self._chk(self.L.nx_prove_machine(self.ctx, self._comps(comps), len(comps), C.byref(cfg), C.c_uint64(seed), adb, C.c_size_t(len(ad)),
                                          C.byref(comm) if comm is not None else None, C.byref(words), C.byref(n), C.byref(stats) if want_stats else None))
        out = np.ctypeslib.as_array(words, shape=(n.value,)).copy()
        self.L.nx_free_host(words)
        return (out, stats.as_dict()) if want_stats else out

    def prove_machine_host(self, comps, cfg, pre_cols, main_cols, ad=b"", coset_order=False, want_stats=False):
        """nx_prove_machine_host: the machine's preprocessed / main traces from HOST memory (lists of contiguous uint32 arrays, component
        after component), uploaded under the commits' own transforms.  Same proof as prove_machine for the same trace."""
        cfg = cfg or default_config()
        keep = [np.ascontiguousarray(c, dtype=np.uint32) for c in list(pre_cols) + list(main_cols)]
        n_pre = len(pre_cols)
        pp = (C.c_void_p * max(1, n_pre))(*[c.ctypes.data for c in keep[:n_pre]])
        mp = (C.c_void_p * max(1, len(keep) - n_pre))(*[c.ctypes.data for c in keep[n_pre:]])
        words, n = C.POINTER(C.c_uint32)(), C.c_size_t(0)
        stats = ProveStats()
        adb = (C.c_uint8 * max(1, len(ad)))(*ad)
        self._chk(self.L.nx_prove_machine_host(self.ctx, self._comps(comps), len(comps), C.byref(cfg), pp, mp, int(bool(coset_order)), adb, C.c_size_t(len(ad)),
                                               C.byref(words), C.byref(n), C.byref(stats) if want_stats else None))
        out = np.ctypeslib.as_array(words, shape=(n.value,)).copy()
        self.L.nx_free_host(words)
        return (out, stats.as_dict()) if want_stats else out

    def machine_claimed_sums(self):
        """`Proof.claimed_sum` of the last prove_machine on this context (reference machine.rs:93-98): (n_components, 4) uint32."""
        n = C.c_uint32(0)
        self._chk(self.L.nx_machine_claimed_sums(self.ctx, None, 0, C.byref(n)))
        out = np.zeros((n.value, 4), np.uint32)
        if n.value:
            self._chk(self.L.nx_machine_claimed_sums(self.ctx, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def prove_sharded(self, comps, comm, cfg=None, seed=1, ad=b"", want_stats=False):
        """One proof, columns sharded over the ranks of `comm` (an NxComm from make_comm); every rank calls this with the
        same arguments and gets the same proof, bit-identical to `prove` on one GPU (nx_prove_synth_sharded)."""
        cfg = cfg or default_config()
        words, n = C.POINTER(C.c_uint32)(), C.c_size_t(0)
        stats = ProveStats()
        adb = (C.c_uint8 * max(1, len(ad)))(*ad)
        self._chk(self.L.nx_prove_synth_sharded(self.ctx, self._comps(comps), len(comps), C.byref(cfg), C.c_uint64(seed), adb, C.c_size_t(len(ad)),
                                                C.byref(comm), C.byref(words), C.byref(n), C.byref(stats) if want_stats else None))
        out = np.ctypeslib.as_array(words, shape=(n.value,)).copy()
        self.L.nx_free_host(words)
        return (out, stats.as_dict()) if want_stats else out
