"""Column-sharded commitment of one trace over the GPUs of a node (BASELINE config #4).

The commit phase of the reference prover (prover/src/machine.rs:208-263: TreeBuilder::extend_evals -> commit) shards by
columns: interpolation and LDE are independent per column (no communication); the Merkle leaf of a row is ONE
sequential Blake2s chain over all columns of the tree, so the shards of a tree form a ring in column order: rank r
continues the 32-byte chaining state of a chunk of rows received from rank r-1 over its own 16-column-aligned block
of columns and forwards it to rank r+1 (32 B per row per hop — the traffic of 8 columns, instead of re-sharding the
LDE by rows).  The last rank finalises the leaves, builds the inner layers and broadcasts the root, so the proof keeps
Stwo's one-tree-per-interaction format and the root is bit-identical to the single-GPU commit.  Row chunks pipeline
the ring: rank r hashes chunk j+1 while rank r+1 hashes chunk j.

One process per GPU; `torch.distributed` is the transport ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  The compute goes through an ops object: `HipShardOps` (libnexus_hip.so) in production; the CPU tests plug in
an oracle-backed object to check the protocol itself (tests/test_sharded_cpu.py).
"""
import numpy as np


def plan_column_shards(n_cols, world):
    """Contiguous column ranges, one per rank, every boundary a multiple of 16 columns (one Blake2s block), balanced
    in units of 16-column blocks; trailing ranks may be empty when there are fewer blocks than ranks."""
    blocks = (n_cols + 15) // 16
    per, extra = divmod(blocks, world)
    out, b = [], 0
    for r in range(world):
        nb = per + (1 if r < extra else 0)
        lo, hi = min(n_cols, 16 * b), min(n_cols, 16 * (b + nb))
        out.append((lo, hi))
        b += nb
    return out


class TorchComm:
    """Point-to-point + broadcast of uint32 buffers over torch.distributed (int32 views: RCCL has no uint32 dtype issue
    for raw moves).  Works with backend 'nccl' (device tensors) and 'gloo' (CPU tensors)."""

    def __init__(self, device):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.device = torch, dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def empty_state(self, n_rows):
        return self.torch.empty((n_rows, 8), dtype=self.torch.int32, device=self.device)

    def send(self, t, dst):
        self.dist.send(t, dst)

    def recv(self, t, src):
        self.dist.recv(t, src)

    def broadcast(self, t, src):
        self.dist.broadcast(t, src)
        return t

    def all_gather(self, out_list, t):
        self.dist.all_gather(out_list, t)

    def device_sync(self):
        if self.device.type == "cuda":
            self.torch.cuda.synchronize(self.device)


class HipShardOps:
    """Shard compute on one MI355X through the C ABI.  State buffers are torch device tensors (plumbing only)."""

    def __init__(self, backend, twiddles):
        self.be, self.tw = backend, twiddles

    def lde(self, cols, log_blowup):
        return self.be.lde(self.tw, cols, log_blowup)          # cols become coefficients, result = LDE columns

    def leaf_chain(self, lde, col_offset, total_cols, state_in, state_out, row_begin, n_rows):
        self.be.merkle_leaf_chain(lde, col_offset, total_cols, state_in.data_ptr() if state_in is not None else None,
                                  state_out.data_ptr(), row_begin, n_rows)
        self.be.sync()                                           # the state leaves this context's stream

    def root_from_leaves(self, leaves, log_size):
        tree = self.be.merkle_from_leaves(leaves.data_ptr(), log_size)
        return tree, tree.root()

    # transposed commit: rows [rb, re) of every local column as one torch tensor / a received block as columns
    def export_rows(self, lde, rb, re):
        import torch, ctypes as C
        n = lde.n_cols
        t = torch.empty((n, re - rb), dtype=torch.int32, device="cuda:%d" % self.be.device)
        for k in range(n):
            self.be._chk(self.be.L.nx_copy(self.be.ctx, C.c_void_p(t.data_ptr() + k * 4 * (re - rb)), C.c_void_p(lde.ptr.value + (k << lde.log_size) * 4 + 4 * rb),
                                           C.c_size_t(re - rb)))
        self.be.sync()
        return t

    def import_block(self, block):
        from . import DeviceColumns
        v = DeviceColumns.view(self.be, block.data_ptr(), block.shape[0], int(np.log2(block.shape[1])))
        v.keep = block                                           # the tensor owns the memory
        return v


def sharded_commit(ops, comm, local_lde, col_range, total_cols, lde_log_size, n_row_chunks=8):
    """Commit one tree whose (equally sized) LDE columns are sharded by `plan_column_shards`.
    local_lde: this rank's LDE columns (object understood by `ops`), col_range = (begin, end) of this rank.
    Returns (root as 8 uint32 words on every rank, tree handle on the last non-empty rank else None)."""
    torch = comm.torch
    shards = [s for s in plan_column_shards(total_cols, comm.world)]
    active = [r for r, (lo, hi) in enumerate(shards) if hi > lo]
    rank = comm.rank
    assert shards[rank] == tuple(col_range), "column range does not match plan_column_shards"
    n_rows_total = 1 << lde_log_size
    n_row_chunks = max(1, min(n_row_chunks, n_rows_total))
    bounds = [n_rows_total * j // n_row_chunks for j in range(n_row_chunks + 1)]
    tree = None
    root = torch.zeros(8, dtype=torch.int32, device=comm.device)
    if rank in active:
        pos = active.index(rank)
        prev_rank = active[pos - 1] if pos > 0 else None
        next_rank = active[pos + 1] if pos + 1 < len(active) else None
        lo, hi = shards[rank]
        leaves = comm.empty_state(n_rows_total) if next_rank is None else None
        for j in range(n_row_chunks):
            rb, re = bounds[j], bounds[j + 1]
            if re == rb:
                continue
            st_in = None
            if prev_rank is not None:
                st_in = comm.empty_state(re - rb)
                comm.recv(st_in, prev_rank)
                comm.device_sync()
            st_out = leaves[rb:re] if leaves is not None else comm.empty_state(re - rb)
            ops.leaf_chain(local_lde, lo, total_cols, st_in, st_out, rb, re - rb)
            if next_rank is not None:
                comm.send(st_out, next_rank)
        if next_rank is None:
            tree, r = ops.root_from_leaves(leaves, lde_log_size)
            root.copy_(torch.from_numpy(np.asarray(r, dtype=np.uint32).view(np.int32)).to(comm.device))
    comm.broadcast(root, active[-1])
    comm.device_sync()
    return root.cpu().numpy().view(np.uint32).copy(), tree


# ---------------------------------------------------------------- the transposed commit (row shards after one all-to-all) ---
# The chaining-state ring moves 32 B per row over every hop and serialises the hashing of a row over the ranks (DESIGN.md §7:
# at config #4 it is the term that does not scale).  The alternative keeps ONLY the LDE column-parallel and then transposes once:
# an all-to-all turns column shards into row shards — rank s receives rows [s M/W, (s+1) M/W) of every column — after which the
# leaf hashing and the whole Merkle subtree over those rows are local; the W subtree roots are all-gathered and every rank
# computes the top log2(W) levels itself ("all-gather only for Merkle roots", BASELINE north star).  Same tree, same root.

def exchange_all_to_all(comm, send):
    """send[s]: tensor for rank s (None = nothing).  Returns recv[q]: the tensor rank q addressed to this rank.  Pairwise
    isend/irecv, so it runs on gloo (CPU tests) and nccl alike; shapes are (n_cols_of_sender, rows_per_rank), known to both sides."""
    dist, torch = comm.dist, comm.torch
    recv = [None] * comm.world
    recv[comm.rank] = send[comm.rank]
    for step in range(1, comm.world):
        dst, src = (comm.rank + step) % comm.world, (comm.rank - step) % comm.world
        ops = []
        if send[dst] is not None and send[dst].numel():
            ops.append(dist.P2POp(dist.isend, send[dst], dst))
        shape = comm.recv_shape(src)
        if shape[0]:
            recv[src] = torch.empty(shape, dtype=torch.int32, device=comm.device)
            ops.append(dist.P2POp(dist.irecv, recv[src], src))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
    return recv


def transposed_commit(ops, comm, local_lde, col_ranges, lde_log_size, exchange=exchange_all_to_all):
    """Commit one tree of equally sized LDE columns sharded by columns (col_ranges[r] = (begin, end) of rank r, any partition):
    all-to-all to row shards, local leaf hashing + subtree, all-gather of the subtree roots, replicated top.  The world size must
    be a power of two not larger than the column length.  Returns (root as 8 uint32 words, local subtree handle)."""
    torch = comm.torch
    W, rank = comm.world, comm.rank
    assert W & (W - 1) == 0 and W <= (1 << lde_log_size)
    log_w = W.bit_length() - 1
    rows = (1 << lde_log_size) // W                       # rows per rank; a contiguous block in the (bit-reversed) row order
    total_cols = col_ranges[-1][1]
    lo, hi = col_ranges[rank]
    comm.recv_shape = lambda q: (col_ranges[q][1] - col_ranges[q][0], rows)
    send = [ops.export_rows(local_lde, s * rows, (s + 1) * rows) if hi > lo else None for s in range(W)]
    recv = exchange(comm, send)
    block = torch.cat([recv[q] for q in range(W) if recv[q] is not None and recv[q].shape[0]], dim=0).contiguous()   # [total_cols, rows]
    assert block.shape == (total_cols, rows)
    leaves = comm.empty_state(rows)
    ops.leaf_chain(ops.import_block(block), 0, total_cols, None, leaves, 0, rows)
    subtree, sub_root = ops.root_from_leaves(leaves, lde_log_size - log_w)
    mine = torch.from_numpy(np.asarray(sub_root, dtype=np.uint32).view(np.int32).copy()).to(comm.device)
    roots = [torch.empty(8, dtype=torch.int32, device=comm.device) for _ in range(W)]
    comm.all_gather(roots, mine)
    top = torch.stack(roots).contiguous()                 # the W subtree roots are level log2(W) of the tree
    _, root = ops.root_from_leaves(top, log_w) if log_w else (None, sub_root)
    comm.device_sync()
    return np.asarray(root, dtype=np.uint32).copy(), subtree


# ---------------------------------------------------------------- transports of the full sharded prove (nx_comm) ----------

class ThreadGroup:
    """All ranks in ONE process (one thread per rank, e.g. several contexts on one GPU): the loopback transport used to
    check nx_prove_synth_sharded against the single-GPU proof on real hardware.  Device buffers are plain pointers in one
    address space, so a 'send' is a device-to-device copy by the receiver."""

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.lock = threading.Lock()
        self.mail = {}            # (src, dst) -> (ptr, n_words)
        self.cv = threading.Condition(self.lock)
        self.slots = [None] * world

    def comm(self, rank, backend):
        return _ThreadComm(self, rank, backend)


class _ThreadComm:
    def __init__(self, group, rank, backend):
        self.g, self.rank, self.be = group, rank, backend

    def abort(self):
        """nx_comm.abort: break the group's barrier — every rank waiting in (or later entering) a collective gets BrokenBarrierError,
        its callback returns non-zero and its prove fails with NX_ERR_HIP instead of waiting for this rank for ever."""
        self.g.barrier.abort()
        with self.g.cv:
            self.g.cv.notify_all()

    def send(self, dst, ptr, n):
        g = self.g
        with g.cv:
            g.mail[(self.rank, dst)] = (ptr, n)
            g.cv.notify_all()
            g.cv.wait_for(lambda: (self.rank, dst) not in g.mail)     # the receiver copied it

    def recv(self, src, ptr, n):
        import ctypes as C
        g = self.g
        with g.cv:
            g.cv.wait_for(lambda: (src, self.rank) in g.mail)
            sp, sn = g.mail[(src, self.rank)]
        assert sn == n
        self.be._chk(self.be.L.nx_copy(self.be.ctx, C.c_void_p(ptr), C.c_void_p(sp), C.c_size_t(n)))
        self.be.sync()
        with g.cv:
            del g.mail[(src, self.rank)]
            g.cv.notify_all()

    def allreduce_m31(self, ptr, n):
        import ctypes as C
        g, be = self.g, self.be
        snap = C.c_void_p()
        be._chk(be.L.nx_alloc(be.ctx, C.c_size_t(n), C.byref(snap)))
        be._chk(be.L.nx_copy(be.ctx, snap, C.c_void_p(ptr), C.c_size_t(n)))
        be.sync()
        g.slots[self.rank] = snap.value
        g.barrier.wait()
        for r in range(g.world):
            if r != self.rank:
                be._chk(be.L.nx_m31_add_into(be.ctx, C.c_void_p(ptr), C.c_void_p(g.slots[r]), C.c_size_t(n)))
        be.sync()
        g.barrier.wait()
        be._chk(be.L.nx_free(be.ctx, snap))

    def allgather(self, data):
        g = self.g
        g.slots[self.rank] = data
        g.barrier.wait()
        out = list(g.slots)
        g.barrier.wait()
        return out

    def broadcast(self, data, nbytes, root):
        g = self.g
        if self.rank == root:
            g.slots[root] = data
        g.barrier.wait()
        out = g.slots[root]
        g.barrier.wait()
        return out

    def _copy(self, dst, src, n):
        import ctypes as C
        if n:
            self.be._chk(self.be.L.nx_copy(self.be.ctx, C.c_void_p(dst), C.c_void_p(src), C.c_size_t(n)))

    def alltoallv(self, send_ptr, soff, scnt, recv_ptr, roff, rcnt):
        g = self.g
        g.slots[self.rank] = (send_ptr, soff, scnt)
        g.barrier.wait()
        for src in range(g.world):
            sp, so, sc = g.slots[src]
            assert sc[self.rank] == rcnt[src], (src, self.rank, sc[self.rank], rcnt[src])
            self._copy(recv_ptr + 4 * roff[src], sp + 4 * so[self.rank], rcnt[src])
        self.be.sync()
        g.barrier.wait()

    def allgather_dev(self, send_ptr, n, recv_ptr):
        g = self.g
        g.slots[self.rank] = send_ptr
        g.barrier.wait()
        for src in range(g.world):
            self._copy(recv_ptr + 4 * n * src, g.slots[src], n)
        self.be.sync()
        g.barrier.wait()


class _DevView:
    """A raw device pointer as a zero-copy torch tensor (int32 words) through __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}


class TorchDistComm:
    """One process per GPU over torch.distributed (backend 'nccl' = RCCL over xGMI).  Device buffers of the library are
    staged through torch tensors (plumbing): the modular all-reduce is a 64-bit sum all-reduce between nx_m31_widen and
    nx_m31_narrow, since RCCL has no modular reduction (8 ranks x (p-1) < 2^34)."""

    def __init__(self, backend, device):
        """device: the torch CUDA device of this rank's context — or None for HOST buffers (the gloo transport tests on CPU: the same
        split / offset logic an 8-GPU RCCL run uses, over plain memory)."""
        import torch
        import torch.distributed as dist
        self.be, self.torch, self.dist, self.device = backend, torch, dist, device
        self.host = device is None or getattr(device, "type", "") == "cpu"
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        # the device collectives run on a stream of their own and the host waits for THAT stream only: the library may have queued
        # the next column chunk's transforms behind the buffer being exchanged (Dist::alltoallv waits for an event, not for its stream)
        self.stream = None if self.host else torch.cuda.Stream(device=device)

    def _sync(self):
        self.be.sync()
        self.torch.cuda.synchronize(self.device)

    def send(self, dst, ptr, n):
        import ctypes as C
        t = self.torch.empty(n, dtype=self.torch.int32, device=self.device)
        self.be._chk(self.be.L.nx_copy(self.be.ctx, C.c_void_p(t.data_ptr()), C.c_void_p(ptr), C.c_size_t(n)))
        self._sync()
        self.dist.send(t, dst)
        self.torch.cuda.synchronize(self.device)

    def recv(self, src, ptr, n):
        import ctypes as C
        t = self.torch.empty(n, dtype=self.torch.int32, device=self.device)
        self.dist.recv(t, src)
        self.torch.cuda.synchronize(self.device)
        self.be._chk(self.be.L.nx_copy(self.be.ctx, C.c_void_p(ptr), C.c_void_p(t.data_ptr()), C.c_size_t(n)))
        self.be.sync()

    def allreduce_m31(self, ptr, n):
        import ctypes as C
        t = self.torch.empty(n, dtype=self.torch.int64, device=self.device)
        self.be._chk(self.be.L.nx_m31_widen(self.be.ctx, C.c_void_p(t.data_ptr()), C.c_void_p(ptr), C.c_size_t(n)))
        self._sync()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        self.torch.cuda.synchronize(self.device)
        self.be._chk(self.be.L.nx_m31_narrow(self.be.ctx, C.c_void_p(ptr), C.c_void_p(t.data_ptr()), C.c_size_t(n)))
        self.be.sync()

    def _bytes_tensor(self, data, nbytes):
        t = self.torch.zeros(max(nbytes, 1), dtype=self.torch.uint8, device=self.device)
        if data is not None and nbytes:
            t[:nbytes] = self.torch.frombuffer(bytearray(data), dtype=self.torch.uint8).to(self.device)
        return t

    def allgather(self, data):
        n = len(data)
        mine = self._bytes_tensor(data, n)
        outs = [self.torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(outs, mine)
        return [bytes(o[:n].cpu().numpy().tobytes()) for o in outs]

    def broadcast(self, data, nbytes, root):
        t = self._bytes_tensor(data, nbytes)
        self.dist.broadcast(t, root)
        return bytes(t[:nbytes].cpu().numpy().tobytes())

    # ---- the device collectives of the row-sharded prove: RCCL works directly on the library's buffers (zero-copy views)
    def _view(self, ptr, n):
        if int(n) == 0:              # a rank without columns hands over a NULL buffer
            return self.torch.empty(0, dtype=self.torch.int32) if self.host else self.torch.empty(0, dtype=self.torch.int32, device=self.device)
        if self.host:                # host memory
            import ctypes as C
            return self.torch.frombuffer((C.c_int32 * max(int(n), 1)).from_address(int(ptr)), dtype=self.torch.int32)[:int(n)]
        return self.torch.as_tensor(_DevView(ptr, max(int(n), 1)), device=self.device)[:int(n)]

    def _to_dev(self, t):
        return t if self.host else t.to(self.device)

    def _dev_sync(self):
        if self.stream is not None:
            self.stream.synchronize()

    def _comm_stream(self):
        import contextlib
        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def alltoallv(self, send_ptr, soff, scnt, recv_ptr, roff, rcnt):
        with self._comm_stream():
            self._alltoallv(send_ptr, soff, scnt, recv_ptr, roff, rcnt)
            self._dev_sync()

    def _alltoallv(self, send_ptr, soff, scnt, recv_ptr, roff, rcnt):
        W = self.world
        contiguous = all(soff[r + 1] == soff[r] + scnt[r] for r in range(W - 1)) and all(roff[r + 1] == roff[r] + rcnt[r] for r in range(W - 1))
        staged = self.dist.get_backend() != "nccl"       # gloo (tests): through host memory
        if contiguous:
            src = self._view(send_ptr + 4 * soff[0], sum(scnt))
            dst = self._view(recv_ptr + 4 * roff[0], sum(rcnt))
            if staged:
                out = self.torch.empty(sum(rcnt), dtype=self.torch.int32)
                self.dist.all_to_all_single(out, src.cpu().contiguous(), list(rcnt), list(scnt))
                dst.copy_(self._to_dev(out))
            else:
                self.dist.all_to_all_single(dst, src, list(rcnt), list(scnt))
        else:
            ops, keep = [], []
            if scnt[self.rank]:      # the own share: a local copy (gloo cannot send to itself)
                assert scnt[self.rank] == rcnt[self.rank]
                self._view(recv_ptr + 4 * roff[self.rank], rcnt[self.rank]).copy_(self._view(send_ptr + 4 * soff[self.rank], scnt[self.rank]))
            for r in range(W):
                if r == self.rank:
                    continue
                if scnt[r]:
                    t = self._view(send_ptr + 4 * soff[r], scnt[r]); t = t.cpu() if staged else t
                    keep.append(t); ops.append(self.dist.P2POp(self.dist.isend, t, r))
                if rcnt[r]:
                    d = self._view(recv_ptr + 4 * roff[r], rcnt[r]); t = self.torch.empty(rcnt[r], dtype=self.torch.int32) if staged else d
                    keep.append((d, t)); ops.append(self.dist.P2POp(self.dist.irecv, t, r))
            for w in self.dist.batch_isend_irecv(ops) if ops else []:
                w.wait()
            if staged:
                for k in keep:
                    if isinstance(k, tuple):
                        k[0].copy_(self._to_dev(k[1]))
        self._dev_sync()

    def allgather_dev(self, send_ptr, n, recv_ptr):
        with self._comm_stream():
            src, dst = self._view(send_ptr, n), self._view(recv_ptr, n * self.world)
            if self.dist.get_backend() != "nccl":
                parts = [self.torch.empty(n, dtype=self.torch.int32) for _ in range(self.world)]
                self.dist.all_gather(parts, src.cpu().contiguous())
                dst.copy_(self._to_dev(self.torch.cat(parts)))
            else:
                self.dist.all_gather_into_tensor(dst, src)
            self._dev_sync()
