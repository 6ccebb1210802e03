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
