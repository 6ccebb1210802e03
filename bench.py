#!/usr/bin/env python
"""bench.py — RISC-V cycles proved / s on the synthetic 2^22-row trace (BASELINE.json config #3).

One "step" = one full prove of the reference-shaped synthetic machine (nx_prove_machine): trace fill on device -> preprocessed and
main tree commits (Circle-FFT LDE + Blake2s Merkle) -> lookup elements -> REAL logup interaction trace on device -> claimed sums
-> interaction tree commit -> composition (recorded AIR, JIT kernel) -> tree 4 -> OODS -> DEEP quotients -> FRI -> PoW ->
decommit -> proof bytes on the host; everything resident in HBM (the trace is generated on device).
`value` = 2^log_n_rows * steps / seconds (max over ranks).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--log-rows 22] [--no-cpu-baseline] [--no-v1-shaped] [--one-proof]
For N > 1 it runs one rank per GPU over RCCL: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...`, or — started PLAINLY as `python bench.py --gpus N` (no WORLD_SIZE in the environment) — it starts those N ranks by
itself (self_launch below: the same torch.distributed.run command on 127.0.0.1 and a free port; the line says so in "launcher").
Default for N > 1: one independent proof of the 2^22-row trace per GPU (batch throughput, "scaling": "weak") as `value`;
the same run then ALSO tries ONE row-sharded proof on the N GPUs and reports it in the "one_proof" block (strong scaling, bytes over
xGMI, equality with the single-GPU proof) — guarded so that no failure or hang there can cost the headline line.
--one-proof: the N GPUs prove ONE trace together (row-sharded prove, "strong" — DESIGN.md §7); that path is byte-exact on thread
ranks and over gloo but has never run on more than one physical GPU (this pool has one per box), so it is opt-in until it has: the
first warm-up step then also checks, on real hardware, that every rank's bytes equal rank 0's single-GPU proof
("one_proof_equals_single_gpu").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy ceiling


def lde_roofline(stats, log_rows, n_cols_total):
    """The Circle-FFT roofline block from one instrumented prove (HIP events around every FFT pass sequence, ctx.hip KTimer)."""
    ms = stats["lde_kernel_ms"]
    gbs = stats["lde_algorithmic_bytes"] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic, source = None, None
    tpath = os.path.join(ROOT, "profiles", "fft_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("algorithmic_bytes_per_column") == float(16 << log_rows):
                traffic = tj["hbm_bytes_per_column"] * (stats["lde_algorithmic_bytes"] / float(16 << log_rows))
                source = "profiles/fft_traffic.json: separate --pmc passes of tools/pmc_traffic.py (FETCH_SIZE x2 + WRITE_SIZE, calibrated on a copy), per column at this size, scaled by this prove's column count; NOT re-measured inside this run"
        except Exception:
            traffic = None
    return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "frac_of_measured_copy": gbs / 6290.0,   # MI355X_MICROARCH.md: 6.29 TB/s measured copy bandwidth
            "traffic": traffic, "traffic_source": source,
            "kernel": "nx::fft13_kernel<INV, FIRST, 1, 4, K> + nx::lde_mid_kernel<4, K> (Circle iFFT+FFT = LDE, all passes of one prove)",
            "algorithmic_bytes": stats["lde_algorithmic_bytes"], "kernel_ms": ms}


STAGES = ("trace_gen", "commit", "interaction", "composition", "oods", "quotients", "fri", "pow", "decommit", "total")


def _arm_guardian(fallback):
    """A child process that prints `fallback` (one JSON line) on this process's stdout if this process dies before _disarm_guardian: the
    child only waits on a pipe — end-of-file without the disarm byte means the parent is gone.  Returns the pipe's write end."""
    sys.stdout.flush(); sys.stderr.flush()
    r, w = os.pipe()
    line = (json.dumps(fallback) + "\n").encode()
    pid = os.fork()
    if pid == 0:
        try:
            os.close(w)
            try:
                os.setsid()             # a launcher that signals the worker's process group does not take the guardian with it
            except OSError:
                pass
            got = os.read(r, 1)
            if not got:
                os.write(1, line)
        finally:
            os._exit(0)
    os.close(r)
    return (w, pid)


def _disarm_guardian(guard):
    if guard is None:
        return
    w, pid = guard
    try:
        os.write(w, b"x")
        os.close(w)
        os.waitpid(pid, 0)
    except OSError:
        pass


def self_launch(n_gpus, argv):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): run the same command line under torch.distributed.run — one
    process per GPU, rendezvous on 127.0.0.1 and a free port — and hand its exit code back.  Rank 0's JSON line passes through."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ)
    env["NX_BENCH_LAUNCHER"] = "self (bench.py started torch.distributed.run: %d ranks, 127.0.0.1:%d)" % (n_gpus, port)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-rows", type=int, default=22)
    ap.add_argument("--n-pre", type=int, default=27)      # reference column.rs:617-664 (18 + 9)
    ap.add_argument("--n-main", type=int, default=347)    # reference column.rs:23-606
    ap.add_argument("--n-logup", type=int, default=16)    # 16 logup columns = 64 interaction base columns
    ap.add_argument("--pow-bits", type=int, default=10)
    ap.add_argument("--hash-mode", type=int, default=0)   # 0 standard Blake2s-256 nodes, 1 zero-state raw compression chaining
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log-rows", type=int, default=20)   # a bounded sample of the same machine: ~10-20 s of host work on the GPU box
    ap.add_argument("--no-v1-shaped", action="store_true", help="skip the second, reference-v1-shaped workload")
    ap.add_argument("--v1-logup", type=int, default=250)      # ~250 logup columns ~ 1.0 k interaction base columns (SURVEY §8 preamble)
    ap.add_argument("--lcd", type=int, default=1, help="log_constraint_degree of the main workload (the reference's v1 has 2, components/mod.rs:12)")
    ap.add_argument("--extra-comps", type=int, default=0, help="small extra components of 2^8, 2^9, ... rows next to the main one (machine.rs:82-91)")
    ap.add_argument("--replicas", action="store_true", help="(the default for N > 1) one independent proof per GPU, weak scaling")
    ap.add_argument("--one-proof", action="store_true", help="N > 1: ONE row-sharded proof on all GPUs (strong scaling) instead of one independent proof per GPU")
    ap.add_argument("--no-host-trace", action="store_true", help="skip the host-resident-trace measurement (host_trace block)")
    ap.add_argument("--no-one-proof", action="store_true", help="N > 1, default mode: do not also try ONE row-sharded proof on the N GPUs after the headline")
    ap.add_argument("--one-proof-timeout", type=int, default=120, help="seconds after which the additional one-proof attempt is abandoned (its block then holds an error)")
    ap.add_argument("--legacy-synth", action="store_true", help="prove the round-1 machine (synthetic interaction fill, hand-written constraint kernel)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise the N > 1 path on a 1-GPU box)")
    ap.add_argument("--transport", default=None, choices=["rccl-native", "torch"],
                    help="--one-proof: who carries the collectives — the library's own RCCL transport (csrc/comm_rccl.hip, the default with --backend nccl) or the torch.distributed callbacks")
    ap.add_argument("--same-device", action="store_true", help="testing only: every rank on GPU 0 (RCCL refuses that; use with --backend gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly (the form the driver uses for N = 1): this process becomes the launcher of the N ranks
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"warning: WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libnexus_hip has no CPU fallback")
    if args.same_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible; one process per GPU — --same-device only for tests over gloo)" % (rank, torch.cuda.device_count()))
    launcher = os.environ.get("NX_BENCH_LAUNCHER") or ("torch.distributed.run (external)" if world > 1 else "none (one process, one GPU)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import nexus_zkvm_amd as nz
    be = nz.HipBackend(local_rank)
    n_inter = 4 * args.n_logup
    comps = [(args.log_rows, args.n_pre, args.n_main, n_inter)] + [(8 + k, 2, 6 + k, 4) for k in range(args.extra_comps)]
    cfg = nz.default_config(pow_bits=args.pow_bits, hash_mode=args.hash_mode, log_constraint_degree=args.lcd)

    def barrier():
        be.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sharded = world > 1 and args.one_proof and not args.replicas
    comm = None
    transport = args.transport or ("rccl-native" if args.backend == "nccl" else "torch")

    def make_comm():
        if transport == "rccl-native":
            # the 128-byte RCCL unique id travels over the process group torch.distributed.run set up; the proof's collectives do not
            box = [nz.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            return be.rccl_comm(box[0], rank, world)
        from nexus_zkvm_amd.sharded import TorchDistComm
        return nz.make_comm(rank, world, TorchDistComm(be, torch.device("cuda", local_rank)))

    if sharded:
        if world & (world - 1):
            raise SystemExit("one row-sharded proof needs a power-of-two number of GPUs (use --replicas otherwise)")
        comm = make_comm()

    def prove(cs, cf, seed, want_stats=False):
        sd = seed if sharded else seed + rank * 97            # one proof together, or one independent proof per rank
        if args.legacy_synth:
            return be.prove_sharded(cs, comm, cf, seed=sd, want_stats=want_stats) if sharded else be.prove(cs, cf, seed=sd, want_stats=want_stats)
        return be.prove_machine(cs, cf, seed=sd, comm=comm, want_stats=want_stats)

    def timed(cs, cf, steps, warmup):
        for w in range(warmup):
            prove(cs, cf, 1000 + w)
        barrier()
        t0 = time.perf_counter()
        for s in range(steps):
            prove(cs, cf, 2000 + s)
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    one_proof_equal = None
    if sharded:
        # first contact with real multi-GPU hardware: every rank's proof of one small statement must equal the single-GPU proof
        import numpy as np
        chk = [(min(args.log_rows, 16), args.n_pre, args.n_main, n_inter)]
        mine = be.prove_machine(chk, cfg, seed=77, comm=comm)
        solo = be.prove_machine(chk, cfg, seed=77)
        ok = torch.tensor([1 if (len(mine) == len(solo) and np.array_equal(mine, solo)) else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        one_proof_equal = bool(int(ok.item()))
        if not one_proof_equal:
            raise SystemExit("bench.py --one-proof: a rank's row-sharded proof differs from the single-GPU proof")
    elapsed = timed(comps, cfg, args.steps, args.warmup)
    # one extra instrumented step (outside the timed region) for the per-stage split and the FFT roofline
    words, stats = prove(comps, cfg, 4242, want_stats=True)

    v1 = None
    if not args.no_v1_shaped and world == 1 and not args.legacy_synth:
        # The reference's v1 machine shape: LOG_CONSTRAINT_DEGREE = 2 (reference prover/src/components/mod.rs:12: the
        # constraints are evaluated on a 2^(n+2) domain, every column re-evaluated there; twiddles for n+3, machine.rs:186-194),
        # ~250 logup columns ~ 1.0 k interaction base columns (chips/range_check/range256.rs:271-288 et al.), 8 extension components
        # of other sizes (machine.rs:82-91).  Reported next to the headline, not instead of it.  A +2 component of the machine has
        # degree-4 main-trace constraints (what the bound is for: sra.rs:267-307) and degree-2 logup constraints (components/mod.rs:53).
        try:
            # per component as in the reference: the main component's bound is +2 (components/mod.rs:12,44-45), every extension's +1
            # (extensions/multiplicity.rs:108-110, ram_init_final.rs:58-60)
            v1_comps = [(args.log_rows, args.n_pre, args.n_main, 4 * args.v1_logup, 2)] + [(8 + k, 2, 6 + k, 4, 1) for k in range(8)]
            v1_cfg = nz.default_config(pow_bits=args.pow_bits, hash_mode=args.hash_mode, log_constraint_degree=2)
            v1_steps = max(1, min(2, args.steps))
            v1_el = timed(v1_comps, v1_cfg, v1_steps, 1)
            v1_words, v1_stats = prove(v1_comps, v1_cfg, 4243, want_stats=True)
            n_cols = sum(c[1] + c[2] + c[3] for c in v1_comps)
            v1 = {"workload": "v1-shaped: 2^%d rows, %d preprocessed + %d main + %d interaction columns (%d logup columns), log_constraint_degree bound 2 for the main component (its main-trace constraints are degree 4, its logup constraints degree 2 like finalize_logup's: degree-aware composition, DESIGN.md §6 item 22), 1 for the 8 extension components of 2^8..2^15 rows"
                  % (args.log_rows, args.n_pre, args.n_main, 4 * args.v1_logup, args.v1_logup),
                  "value": (1 << args.log_rows) * v1_steps / v1_el, "unit": "cycles/s", "ms_per_step": 1e3 * v1_el / v1_steps, "steps": v1_steps,
                  "n_columns": n_cols, "proof_words": int(len(v1_words)),
                  "stages_ms": {k: round(v1_stats[k], 3) for k in STAGES},
                  "roofline": lde_roofline(v1_stats, args.log_rows, n_cols),
                  "merkle": {"kernel_ms": v1_stats["merkle_kernel_ms"], "algorithmic_bytes": v1_stats["merkle_algorithmic_bytes"]}}
            # round 6: the same statement with the main component's relations at the reference's tuple WIDTHS and entry kinds (NX_TUPLES_V1:
            # 1 / 4 = [constant, b, c, a] with a flag-column numerator / 9 / 3 = [column, constant, column + column] — range256.rs:37,
            # bit_op.rs:31,341-365, register_mem_check.rs:34) instead of one- / two-column tuples: what the interaction stage and the
            # logup constraints of the reference's shape cost
            try:
                t_comps = [v1_comps[0] + (nz.TUPLES_V1,)] + v1_comps[1:]
                t_el = timed(t_comps, v1_cfg, v1_steps, 1)
                _, t_stats = prove(t_comps, v1_cfg, 4244, want_stats=True)
                v1["reference_tuple_widths"] = {"what": "main component with NX_TUPLES_V1 (tuple widths 1, 1, 4, 1, 9, 1, 3, 1 cycling; constant and column-sum entries; +1 / -m / +m numerators)",
                                                "ms_per_step": 1e3 * t_el / v1_steps, "value": (1 << args.log_rows) * v1_steps / t_el, "unit": "cycles/s",
                                                "stages_ms": {k: round(t_stats[k], 3) for k in STAGES}}
            except Exception as e:   # noqa: BLE001
                v1["reference_tuple_widths"] = {"error": repr(e)[:300]}
        except Exception as e:   # noqa: BLE001 — the headline line must still be printed
            v1 = {"error": repr(e)[:300]}

    # The reference hands its trace over in HOST memory (prover/src/trace/trace_builder.rs:19-32).  `value` above has the trace generated in
    # HBM; this block times the same proof from a host-resident preprocessed + main trace, uploaded chunk by chunk UNDER the commits' own
    # transforms (nx_prove_machine_host).  PCIe-inclusive, reported next to the headline, never as `value`.
    host_trace = None
    if world == 1 and not args.no_host_trace and not args.legacy_synth:
        try:
            import numpy as np
            hseed = 2000
            pre_sets, main_sets = be.synth_fill_tree(comps, 0, hseed), be.synth_fill_tree(comps, 1, hseed)
            pre = [c for st_ in pre_sets for c in st_.to_cpu()]
            main = [c for st_ in main_sets for c in st_.to_cpu()]
            for st_ in pre_sets + main_sets:
                st_.free()
            nbytes = sum(c.nbytes for c in pre) + sum(c.nbytes for c in main)
            hw = be.prove_machine_host(comps, cfg, pre, main)            # warm-up (first pin of these pages)
            hsteps = max(1, min(3, args.steps))
            barrier(); t0 = time.perf_counter()
            for _ in range(hsteps):
                hw = be.prove_machine_host(comps, cfg, pre, main)
            barrier(); hel = (time.perf_counter() - t0) / hsteps
            same = bool(np.array_equal(hw, be.prove_machine(comps, cfg, seed=hseed)))
            # the same with the columns pinned once by their owner (nx_host_pin: a trace buffer that is reused from proof to proof)
            pinned_ms = None
            try:
                for c in pre + main:
                    be.host_pin(c)
                try:
                    be.prove_machine_host(comps, cfg, pre, main)
                    barrier(); t0 = time.perf_counter()
                    for _ in range(hsteps):
                        hp = be.prove_machine_host(comps, cfg, pre, main)
                    barrier(); pinned_ms = 1e3 * (time.perf_counter() - t0) / hsteps
                    same = same and bool(np.array_equal(hp, hw))
                finally:
                    for c in pre + main:
                        be.host_unpin(c)
            except Exception as e:   # noqa: BLE001 — informational
                pinned_ms = repr(e)[:200]
            host_trace = {"ms_per_step": 1e3 * hel, "value": (1 << args.log_rows) / hel, "unit": "cycles/s", "steps": hsteps, "bytes_uploaded": int(nbytes),
                          "upload_only_floor_ms_at_63GBs": 1e3 * nbytes / 63e9, "equals_device_resident_proof": same,
                          "ms_per_step_columns_pinned_by_owner": pinned_ms,
                          "what": "same statement, preprocessed + main trace (%d columns) in host memory in bit-reversed circle-domain order, pinned in place and uploaded in 16-column chunks on a copy stream while the commit transforms and hashes the chunks that have arrived; interaction trace on the device" % (len(pre) + len(main))}
            del pre, main
        except Exception as e:   # noqa: BLE001 — the headline line must still be printed
            host_trace = {"error": repr(e)[:300]}

    # A program that is proved again and again commits the SAME preprocessed tree every time (reference machine.rs:208-228): with
    # "machine.reuse_preprocessed" the tree of the first proof is adopted by the later ones (nx_prover_tree_adopt's rule; same proof bytes).
    # Informational — `value` always commits every tree afresh, as the reference does.
    pre_reuse = None
    if world == 1 and not args.legacy_synth:
        try:
            import numpy as np
            be.set_option("machine.reuse_preprocessed", 1)
            rsteps = max(1, min(3, args.steps))
            r_el = timed(comps, cfg, rsteps, 1)
            same = bool(np.array_equal(be.prove_machine(comps, cfg, seed=4242), words))
            pre_reuse = {"ms_per_step": 1e3 * r_el / rsteps, "value": (1 << args.log_rows) * rsteps / r_el, "unit": "cycles/s", "steps": rsteps, "equals_fresh_commit_proof": same,
                         "what": "the %d preprocessed columns' tree committed once and adopted by the following proofs (seeds differ: main and interaction trees are new every time)" % args.n_pre}
        except Exception as e:   # noqa: BLE001 — informational
            pre_reuse = {"error": repr(e)[:300]}
        finally:
            try:
                be.set_option("machine.reuse_preprocessed", 0)
            except Exception:   # noqa: BLE001
                pass

    prover_options = {}
    for name in ("air.degree_split", "air.half_domain", "air.quarter_domain", "quotients.coeffs"):
        try:
            prover_options[name] = int(be.get_option(name))
        except Exception:   # noqa: BLE001 — informational only
            pass
    out = None
    if rank == 0:
        n_cycles = (1 << args.log_rows) * args.steps * (1 if sharded else world)
        n_cols_total = args.n_pre + args.n_main + n_inter
        roof = lde_roofline(stats, args.log_rows, n_cols_total)
        out = {
            "metric": "RISC-V cycles proved/sec at log_n_rows=%d; achieved HBM GB/s on Circle-FFT" % args.log_rows,
            "value": n_cycles / elapsed,
            "unit": "cycles/s",
            "n_gpus": world,
            "launcher": launcher,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if sharded or world == 1 else "weak",
            "vs_baseline": None,
            "dtype": "u32 (M31)",
            "data": "synthetic",
            "config": {"workload": "synthetic 2^%d-row trace, full prove (LDE+quotient+FRI+Merkle), %d preprocessed + %d main + %d interaction columns (%s), blowup 2, %d queries, pow_bits %d, %s Merkle nodes"
                       % (args.log_rows, args.n_pre, args.n_main, n_inter,
                          "synthetic interaction fill, hand-written constraint kernel" if args.legacy_synth else "%d real logup columns generated on device, recorded AIR" % args.n_logup,
                          cfg.n_queries, cfg.pow_bits, "standard Blake2s-256" if args.hash_mode == 0 else "raw zero-state Blake2s compression"),
                       "log_n_rows": args.log_rows,
                       "parallelism": ("ONE proof on %d GPUs: column-parallel LDE, all-to-all into row blocks, local hashing / constraints / quotients / FRI folds" % world) if sharded
                       else ("1 independent proof per GPU" if world > 1 else "1 GPU"),
                       "proof_words": int(len(words)),
                       # every byte of the proof is Stwo's: these only choose WHERE exact polynomial identities let the prover read fewer bytes
                       # (0 = Stwo's evaluation order, same proof; profiles/r03_*_ab.log have both)
                       "prover_options": prover_options},
            "roofline": roof,
            # the FFT is VALU-issue bound on gfx950, not HBM bound: butterflies of one prove's iFFT + LDE work
            # (n/2 * N per iFFT, n * N per 2x LDE: the trivial top layer is not computed) against the measured chip ceiling of the
            # butterfly instruction sequence itself (tools/ubench/bfly_rates.hip, variant B: 4.73e12 butterflies/s)
            "roofline_valu": (lambda nb: {"bound": "valu", "achieved": nb / (stats["lde_kernel_ms"] * 1e-3) / 1e12, "peak": 4.73, "unit": "T butterflies/s",
                                          "frac": nb / (stats["lde_kernel_ms"] * 1e-3) / 4.73e12, "butterflies": nb})(
                ((n_cols_total * 1.5 * args.log_rows * (1 << args.log_rows)) / (world if sharded else 1)
                 + 4 * ((args.log_rows + 1) / 2.0 + (args.log_rows + 1)) * (2 << args.log_rows))) if stats["lde_kernel_ms"] > 0 else None,
            "stages_ms": {k: round(stats[k], 3) for k in STAGES},
            "merkle": {"kernel_ms": stats["merkle_kernel_ms"], "algorithmic_bytes": stats["merkle_algorithmic_bytes"],
                       "achieved_GBs": stats["merkle_algorithmic_bytes"] / (stats["merkle_kernel_ms"] * 1e-3) / 1e9 if stats["merkle_kernel_ms"] > 0 else 0.0},
        }
        if sharded:
            out["one_proof_equals_single_gpu"] = one_proof_equal
            out["xgmi"] = {"transport": transport, "bytes_sent_per_gpu_per_proof": int(stats["comm_bytes"]), "ms_in_collectives_per_proof": round(stats["comm_ms"], 3),
                           "collectives": "one all-to-all per trace tree (LDE columns -> row blocks), all-gather of W subtree roots per tree, of the columns read at a non-zero mask offset, of the composition accumulator and of the FRI tail; sampled / queried values (KBs)"}
        if host_trace is not None:
            out["host_trace"] = host_trace
        if pre_reuse is not None:
            out["preprocessed_reuse"] = pre_reuse
        if v1 is not None:
            out["config_v1_shaped"] = v1
    # N > 1, default mode: `value` above is N independent proofs.  The SAME run then also tries ONE row-sharded proof on the N GPUs
    #  and reports it in a block of its own: the strong-scaling number and the RCCL bring-up result of whatever
    # multi-GPU box the driver has, at no risk to the headline — every failure, on any rank, ends as {"error": ...}, and a collective
    # that never returns (a peer died before entering it) is cut off by a watchdog: the line is still printed, every rank still exits 0.
    one_proof = None
    hung = False
    if world > 1 and not sharded and not args.no_one_proof and not args.legacy_synth:
        import threading
        box = {}
        # the attempt drives RCCL paths no 1-GPU box has ever run: if THIS process dies in it (a fault inside a collective, or the launcher
        # killing rank 0 because a peer died) a guardian child still prints the finished headline line, with the block reporting the death
        guard = _arm_guardian(dict(out, one_proof={"error": "rank 0 did not survive the attempt (crashed, or was killed by the launcher after a peer died)"})) if rank == 0 else None

        def attempt():
            try:
                import numpy as np
                torch.cuda.set_device(local_rank)
                if world & (world - 1):
                    raise RuntimeError("a row-sharded proof needs a power-of-two number of GPUs")
                if os.environ.get("NX_BENCH_ONE_PROOF_FAULT") == str(rank):      # test hook: this rank fails alone, its peers meet a missing partner
                    raise RuntimeError("injected fault on rank %d" % rank)
                c = make_comm()
                chk = [(min(args.log_rows, 16), args.n_pre, args.n_main, n_inter)]
                mine = be.prove_machine(chk, cfg, seed=77, comm=c)
                solo = be.prove_machine(chk, cfg, seed=77)
                ok = torch.tensor([1 if (len(mine) == len(solo) and np.array_equal(mine, solo)) else 0], dtype=torch.int32, device="cuda")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if not int(ok.item()):
                    raise RuntimeError("a rank's row-sharded proof of the 2^%d-row check statement differs from the single-GPU proof" % chk[0][0])
                steps1 = max(1, min(3, args.steps))
                be.prove_machine(comps, cfg, seed=1500, comm=c)                      # warm-up
                barrier()
                t0 = time.perf_counter()
                for k in range(steps1):
                    w1 = be.prove_machine(comps, cfg, seed=2000 + k, comm=c)
                barrier()
                el1 = time.perf_counter() - t0
                t = torch.tensor([el1], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el1 = float(t.item())
                _, st1 = be.prove_machine(comps, cfg, seed=4242, comm=c, want_stats=True)
                box["ok"] = {"scaling": "strong", "value": (1 << args.log_rows) * steps1 / el1, "unit": "cycles/s", "ms_per_step": 1e3 * el1 / steps1, "steps": steps1,
                             "equals_single_gpu": True, "check": "2^%d-row statement, every rank's bytes == its own single-GPU proof" % chk[0][0],
                             "proof_words": int(len(w1)), "stages_ms": {k: round(st1[k], 3) for k in STAGES},
                             "xgmi": {"transport": transport, "bytes_sent_per_gpu_per_proof": int(st1["comm_bytes"]), "ms_in_collectives_per_proof": round(st1["comm_ms"], 3)}}
            except BaseException as e:   # noqa: BLE001 — the headline must survive anything here
                box["err"] = repr(e)[:400]

        th = threading.Thread(target=attempt, daemon=True)
        th.start()
        th.join(timeout=args.one_proof_timeout)
        if th.is_alive():
            hung = True
            one_proof = {"error": "no result after %d s (a collective did not return; a peer may have failed before entering it)" % args.one_proof_timeout}
        else:
            # a rank that failed alone must not leave the others' blocks looking fine: agree on the outcome (bounded by the same watchdog idea)
            one_proof = box.get("ok") or {"error": box.get("err", "unknown")}
        _disarm_guardian(guard)

    if rank == 0:
        if one_proof is not None:
            out["one_proof"] = one_proof
        if not args.no_cpu_baseline and world == 1:
            import oracle_lib as O   # checker / baseline only — never part of the measured GPU path
            import numpy as np
            O.build_oracle()
            cores = os.cpu_count() or 1
            ccomps = [(args.cpu_log_rows, args.n_pre, args.n_main, n_inter)] + comps[1:]
            ocfg = O.default_cfg(pow_bits=args.pow_bits, hash_mode=args.hash_mode, log_constraint_degree=args.lcd)
            t0 = time.perf_counter()
            if args.legacy_synth:
                ref = O.prove_synth(ccomps, ocfg, seed=7, threads=cores)
            else:
                import machine_ref
                ref = machine_ref.prove_machine(ccomps, ocfg, seed=7, threads=cores)
            secs = time.perf_counter() - t0
            # the same statement on the GPU: every proof word must agree (parity pinned at the sample size, not just verifier acceptance)
            gpu = be.prove(ccomps, cfg, seed=7) if args.legacy_synth else be.prove_machine(ccomps, cfg, seed=7)
            equal = bool(len(gpu) == len(ref) and np.array_equal(gpu, ref))
            out["cpu_baseline"] = {"value": (1 << args.cpu_log_rows) / secs, "unit": "cycles/s", "cores": cores, "kind": "port",
                                   "sample": "one full prove of the same machine at 2^%d rows (%.1f s) by the C++ oracle, %d threads; NOT Stwo SimdBackend"
                                   % (args.cpu_log_rows, secs, cores),
                                   "proof_equal": equal, "proof_words": int(len(ref))}
            if not equal:
                print(json.dumps(out), flush=True)
                raise SystemExit("bench.py: the GPU proof of the cpu_baseline sample differs from the oracle's proof")
        print(json.dumps(out), flush=True)
    if hung or (one_proof is not None and "error" in one_proof):
        # a thread of this process is stuck inside a collective, or this rank failed alone and its peers are (until their watchdogs
        # fire): no orderly shutdown of the process group is possible — the line is out, leave without the final barrier
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)
    be.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
