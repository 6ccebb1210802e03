"""world_size > 1 tests of the column-sharded commitment protocol (nexus-zkvm_amd/sharded.py) on CPU: gloo transport,
the CPU oracle as the per-rank compute (test infrastructure only).  What is checked is the PROTOCOL: shard plan,
chaining-state ring over row chunks, finalisation on the last shard, root broadcast — the root every rank ends with
must be the root of the oracle's single-process commit of all columns."""
import os
import socket

import numpy as np
import pytest

import oracle_lib as O

P = O.P


def test_plan_column_shards():
    from nexus_zkvm_amd.sharded import plan_column_shards
    for n_cols in (1, 4, 15, 16, 17, 27, 64, 347, 1388):
        for world in (1, 2, 3, 4, 8):
            sh = plan_column_shards(n_cols, world)
            assert len(sh) == world and sh[0][0] == 0 and sh[-1][1] == n_cols
            for (a, b), (c, d) in zip(sh, sh[1:]):
                assert b == c and a <= b
            for lo, hi in sh:
                assert lo % 16 == 0 or lo == n_cols
            sizes = [(hi - lo + 15) // 16 for lo, hi in sh]
            assert max(sizes) - min(sizes) <= 1
    assert plan_column_shards(347, 8) == [(0, 48), (48, 96), (96, 144), (144, 192), (192, 240), (240, 288), (288, 320), (320, 347)]


class OracleShardOps:
    """Same interface as nexus_zkvm_amd.sharded.HipShardOps, computed by the CPU oracle on numpy arrays."""

    def __init__(self, mode, log_size):
        self.mode = mode
        self.tw = O.Twiddles(log_size + 1)

    def lde(self, cols, log_blowup):
        log = int(np.log2(cols.shape[1]))
        return np.stack([self.tw.evaluate(self.tw.interpolate(c), log + log_blowup) for c in cols])

    def leaf_chain(self, lde, col_offset, total_cols, state_in, state_out, row_begin, n_rows):
        L = O.lib()
        n_cols = lde.shape[0]
        last_shard = col_offset + n_cols == total_cols
        iv = np.array([0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19], np.uint32)
        for r in range(n_rows):
            if state_in is not None:
                h = state_in[r].numpy().view(np.uint32).copy()
            elif self.mode == O.HASH_STD:
                h = iv.copy(); h[0] ^= 0x01010020
            else:
                h = np.zeros(8, np.uint32)
            t = 4 * col_offset
            for c0 in range(0, n_cols, 16):
                m = np.zeros(16, np.uint32)
                k = min(16, n_cols - c0)
                m[:k] = lde[c0:c0 + k, row_begin + r]
                fin = last_shard and c0 + 16 >= n_cols
                if self.mode == O.HASH_STD:
                    t = 4 * total_cols if fin else t + 64
                    L.orc_blake2s_compress(O.ptr(h), O.ptr(m), t, 0, 0xFFFFFFFF if fin else 0, 0)
                else:
                    L.orc_blake2s_compress(O.ptr(h), O.ptr(m), 0, 0, 0, 0)
            state_out[r] = __import__("torch").from_numpy(h.view(np.int32))

    def export_rows(self, lde, rb, re):
        return __import__("torch").from_numpy(np.ascontiguousarray(lde[:, rb:re]).view(np.int32).copy())

    def import_block(self, block):
        return block.numpy().view(np.uint32)

    def root_from_leaves(self, leaves, log_size):
        L = O.lib()
        layer = leaves.numpy().view(np.uint32).copy()
        for _ in range(log_size):
            nxt = np.zeros((layer.shape[0] // 2, 8), np.uint32)
            for i in range(nxt.shape[0]):
                ch = np.ascontiguousarray(layer[2 * i:2 * i + 2].reshape(-1))
                out = np.zeros(8, np.uint32)
                L.orc_hash_node(O.ptr(ch), None, 0, self.mode, O.ptr(out))
                nxt[i] = out
            layer = nxt
        return None, layer[0]


def _worker(rank, world, port, n_cols, log_size, mode, n_chunks, result_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from nexus_zkvm_amd.sharded import TorchComm, plan_column_shards, sharded_commit
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = TorchComm(torch.device("cpu"))
        cols = np.random.default_rng(1234).integers(0, P, (n_cols, 1 << log_size), dtype=np.uint32)   # same on every rank
        lo, hi = plan_column_shards(n_cols, world)[rank]
        ops = OracleShardOps(mode, log_size)
        local = ops.lde(cols[lo:hi], 1) if hi > lo else np.zeros((0, 2 << log_size), np.uint32)
        root, _ = sharded_commit(ops, comm, local, (lo, hi), n_cols, log_size + 1, n_row_chunks=n_chunks)
        np.save(os.path.join(result_dir, f"root_{rank}.npy"), root)
        if rank == 0:
            full = ops.lde(cols, 1)
            np.save(os.path.join(result_dir, "expected.npy"), O.merkle_commit(list(full), mode))
    finally:
        dist.destroy_process_group()


def _transposed_worker(rank, world, port, n_cols, log_size, mode, uneven, result_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from nexus_zkvm_amd.sharded import TorchComm, transposed_commit
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = TorchComm(torch.device("cpu"))
        cols = np.random.default_rng(4321).integers(0, P, (n_cols, 1 << log_size), dtype=np.uint32)   # same on every rank
        # any column partition works (no 16-column alignment: the leaves are hashed whole after the transposition)
        cuts = [0] + [min(n_cols, (n_cols * (r + 1)) // world + (3 if uneven and r == 0 else 0)) for r in range(world - 1)] + [n_cols]
        if uneven and world > 2:
            cuts[2] = cuts[1]                                      # a rank without columns
        ranges = [(cuts[r], cuts[r + 1]) for r in range(world)]
        lo, hi = ranges[rank]
        ops = OracleShardOps(mode, log_size)
        local = ops.lde(cols[lo:hi], 1) if hi > lo else np.zeros((0, 2 << log_size), np.uint32)
        root, _ = transposed_commit(ops, comm, local, ranges, log_size + 1)
        np.save(os.path.join(result_dir, f"root_{rank}.npy"), root)
        if rank == 0:
            np.save(os.path.join(result_dir, "expected.npy"), O.merkle_commit(list(ops.lde(cols, 1)), mode))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_cols,mode,uneven", [(2, 21, O.HASH_STD, False), (4, 37, O.HASH_STD, True), (2, 16, O.HASH_RAW0, True), (4, 5, O.HASH_RAW0, False)])
def test_transposed_commit_matches_single_process_root(tmp_path, world, n_cols, mode, uneven):
    """Column-parallel LDE, one all-to-all to row shards, local leaf hashing and subtrees, all-gather of the subtree roots, replicated
    top: every rank ends with the single-process root (the plan DESIGN.md §7 proposes for 8 GPUs, protocol checked over gloo)."""
    import torch.multiprocessing as mp
    O.build_oracle()
    mp.spawn(_transposed_worker, args=(world, _free_port(), n_cols, 4, mode, uneven, str(tmp_path)), nprocs=world, join=True)
    expected = np.load(tmp_path / "expected.npy")
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"root_{r}.npy"), expected), (world, n_cols, mode, r)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_cols,mode,n_chunks", [(2, 40, O.HASH_STD, 4), (2, 33, O.HASH_RAW0, 3), (3, 20, O.HASH_STD, 1), (2, 16, O.HASH_STD, 2)])
def test_sharded_commit_ring_matches_single_process_root(tmp_path, world, n_cols, mode, n_chunks):
    import torch.multiprocessing as mp
    O.build_oracle()
    mp.spawn(_worker, args=(world, _free_port(), n_cols, 4, mode, n_chunks, str(tmp_path)), nprocs=world, join=True)
    expected = np.load(tmp_path / "expected.npy")
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"root_{r}.npy"), expected), (world, n_cols, mode, r)


def _host_comm_worker(rank, world, port, result_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from nexus_zkvm_amd.sharded import TorchDistComm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = TorchDistComm(None, torch.device("cpu"))     # host-buffer half of the transport needs no GPU
        parts = comm.allgather(bytes([rank + 1]) * 5 + b"x" * rank)[:world] if False else comm.allgather(bytes([rank + 1]) * 6)
        assert parts == [bytes([r + 1]) * 6 for r in range(world)]
        for root in range(world):
            got = comm.broadcast(b"root%d!" % root if rank == root else None, 6, root)
            assert got == b"root%d!" % root
        assert comm.broadcast(None if rank else b"", 0, 0) == b""
        open(os.path.join(result_dir, f"ok_{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_torch_transport_host_buffers_over_gloo(tmp_path):
    """allgather / broadcast of nexus_zkvm_amd.sharded.TorchDistComm (sampled values, queried values, roots, witnesses of the
    sharded prove) with world_size 3 over gloo; the device-buffer half (ring, modular all-reduce) is covered on the GPU box."""
    import torch.multiprocessing as mp
    world = 3
    mp.spawn(_host_comm_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok_{r}").exists() for r in range(world))


# ---------------- the row-sharded prove: column plan and the device collectives' split logic (host buffers over gloo) ----------

def test_plan_local_columns_partitions_every_run():
    """nx_plan_local_columns (host arithmetic of libnexus_hip.so, no GPU): consecutive groups of one size form a run; the ranks'
    ranges partition every run contiguously and in rank order, balanced to within one column."""
    import ctypes as C
    import nexus_zkvm_amd as nz
    L = nz.load_library()
    groups = [(27, 12), (5, 12), (0, 12), (9, 10), (3, 12), (1, 7), (2, 7)]
    n = np.array([g[0] for g in groups], np.uint32); lg = np.array([g[1] for g in groups], np.uint32)
    for world in (1, 2, 4, 8):
        owned = [[[] for _ in groups] for _ in range(world)]
        for r in range(world):
            lo, hi = np.zeros(len(groups), np.uint32), np.zeros(len(groups), np.uint32)
            assert L.nx_plan_local_columns(n.ctypes.data_as(C.c_void_p), lg.ctypes.data_as(C.c_void_p), len(groups), r, world, lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p)) == 0
            for g in range(len(groups)):
                assert lo[g] <= hi[g] <= n[g]
                owned[r][g] = list(range(lo[g], hi[g]))
        for g in range(len(groups)):
            assert sum((owned[r][g] for r in range(world)), []) == list(range(n[g])), (world, g)        # a partition, in rank order
        for run in ([0, 1, 2], [3], [4], [5, 6]):
            per_rank = [sum(len(owned[r][g]) for g in run) for r in range(world)]
            assert max(per_rank) - min(per_rank) <= 1, (world, run, per_rank)
    bad = np.zeros(1, np.uint32)
    assert L.nx_plan_local_columns(n.ctypes.data_as(C.c_void_p), lg.ctypes.data_as(C.c_void_p), 1, 3, 2, bad.ctypes.data_as(C.c_void_p), bad.ctypes.data_as(C.c_void_p)) != 0


def _rowshard_worker(rank, world, port, result_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from nexus_zkvm_amd.sharded import TorchDistComm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = TorchDistComm(None, None)                     # HOST buffers: the split / offset logic of the RCCL path, over plain memory
        cut = lambda n, r: n * r // world
        # (1) the transposition of TreeBuilder::commit_dist: column shards [lo, hi) x M rows -> rows [r Mb, (r+1) Mb) of ALL columns
        n_run, M = 11, 64
        mb = M // world
        full = (np.arange(n_run * M, dtype=np.uint32).reshape(n_run, M) * 2654435761 % 1000003).astype(np.uint32)
        lo, hi = cut(n_run, rank), cut(n_run, rank + 1)
        n_loc = hi - lo
        send = np.ascontiguousarray(np.stack([full[lo:hi, d * mb:(d + 1) * mb] for d in range(world)]).reshape(-1) if n_loc else np.zeros(0, np.uint32))   # transpose_blocks(pack)
        recv = np.zeros(n_run * mb, np.uint32)
        soff, scnt = [d * n_loc * mb for d in range(world)], [n_loc * mb] * world
        roff, rcnt = [cut(n_run, s) * mb for s in range(world)], [(cut(n_run, s + 1) - cut(n_run, s)) * mb for s in range(world)]
        comm.alltoallv(send.ctypes.data if n_loc else recv.ctypes.data, soff, scnt, recv.ctypes.data, roff, rcnt)
        assert np.array_equal(recv.reshape(n_run, mb), full[:, rank * mb:(rank + 1) * mb])
        # (2) back: row blocks of all columns -> column shards (the logup interaction trace before its LDE)
        rows = np.ascontiguousarray(full[:, rank * mb:(rank + 1) * mb]).reshape(-1)
        back = np.zeros(max(n_loc, 1) * M, np.uint32)
        comm.alltoallv(rows.ctypes.data, [cut(n_run, d) * mb for d in range(world)], [(cut(n_run, d + 1) - cut(n_run, d)) * mb for d in range(world)],
                       back.ctypes.data, [s * n_loc * mb for s in range(world)], [n_loc * mb] * world)
        if n_loc:
            got = back[:world * n_loc * mb].reshape(world, n_loc, mb).transpose(1, 0, 2).reshape(n_loc, M)             # transpose_blocks(unpack)
            assert np.array_equal(got, full[lo:hi])
        # (3) receive regions that are NOT contiguous in rank order (columns_on_eval_domain with interleaved owners): the point-to-point route
        blocks = np.arange(world * 5, dtype=np.uint32) + 100 * rank
        scat = np.zeros(world * 8, np.uint32)
        comm.alltoallv(blocks.ctypes.data, [5 * d for d in range(world)], [5] * world, scat.ctypes.data, [8 * ((s + 1) % world) for s in range(world)], [5] * world)
        for s_ in range(world):
            assert np.array_equal(scat[8 * ((s_ + 1) % world):8 * ((s_ + 1) % world) + 5], np.arange(5 * rank, 5 * rank + 5, dtype=np.uint32) + 100 * s_)
        # (4) all-gather of a row block into the whole column (masked columns, composition accumulator, FRI tail)
        whole = np.zeros(M, np.uint32)
        mine = np.ascontiguousarray(full[3, rank * mb:(rank + 1) * mb])
        comm.allgather_dev(mine.ctypes.data, mb, whole.ctypes.data)
        assert np.array_equal(whole, full[3])
        open(os.path.join(result_dir, f"ok_{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_row_shard_collectives_over_gloo(tmp_path, world):
    """The two device collectives of the row-sharded prove (nx_comm.alltoallv / allgather_dev as nexus_zkvm_amd.sharded.TorchDistComm
    implements them for RCCL) with world_size 2 and 4 over gloo on host buffers, on the exact offset / count patterns the prover
    produces: the column-shard -> row-block transposition with a column count that does not divide (11 columns), its inverse, a
    receive layout that is not contiguous in rank order, and the row-block all-gather."""
    import torch.multiprocessing as mp
    mp.spawn(_rowshard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok_{r}").exists() for r in range(world))
