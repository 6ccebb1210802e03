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
