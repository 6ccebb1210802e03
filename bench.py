#!/usr/bin/env python
"""bench.py — RISC-V cycles proved / s on the synthetic 2^22-row trace (BASELINE.json config #3).

One "step" = one full prove of the synthetic machine (trace fill on device -> 3 tree commits
(Circle-FFT LDE + Blake2s Merkle) -> composition -> tree 4 -> OODS -> DEEP quotients -> FRI -> PoW ->
decommit -> proof bytes on the host), everything resident in HBM (the trace is generated on device).
`value` = 2^log_n_rows * steps * n_gpus / seconds (max over ranks).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--log-rows 22] [--no-cpu-baseline]
For N > 1 it is launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(one rank per GPU, RCCL); each rank proves its own trace (independent proofs — see DESIGN.md §multi-GPU).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-rows", type=int, default=22)
    ap.add_argument("--n-pre", type=int, default=27)      # reference column.rs:617-664 (18 + 9)
    ap.add_argument("--n-main", type=int, default=347)    # reference column.rs:23-606
    ap.add_argument("--n-inter", type=int, default=64)    # 16 logup columns x 4 base columns
    ap.add_argument("--pow-bits", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log-rows", type=int, default=20)   # a bounded sample of the same machine: ~10 s of host work on the GPU box
    ap.add_argument("--sharded", action="store_true",
                    help="N > 1: ONE proof per step with its columns sharded over the N GPUs (config #4 style, strong scaling) instead of one independent proof per GPU")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libnexus_hip has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import nexus_zkvm_amd as nz
    be = nz.HipBackend(local_rank)
    comps = [(args.log_rows, args.n_pre, args.n_main, args.n_inter)]
    cfg = nz.default_config(pow_bits=args.pow_bits)

    def barrier():
        be.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sharded = args.sharded and world > 1
    if sharded:
        from nexus_zkvm_amd.sharded import TorchDistComm
        comm = nz.make_comm(rank, world, TorchDistComm(be, torch.device("cuda", local_rank)))
        prove = lambda seed: be.prove_sharded(comps, comm, cfg, seed=seed)          # same seed on every rank: one proof
    else:
        prove = lambda seed: be.prove(comps, cfg, seed=seed + rank * 97)            # one independent proof per rank
    for w in range(args.warmup):
        prove(1000 + w)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        prove(2000 + s)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # one extra instrumented step (outside the timed region) for the per-stage split and the FFT roofline:
    # HIP events on the context's stream around every Circle-FFT pass sequence (nexus-zkvm_amd/csrc/ctx.hip KTimer)
    words, stats = be.prove(comps, cfg, seed=4242 + rank, want_stats=True)
    out = None
    if rank == 0:
        n_cycles = (1 << args.log_rows) * args.steps * (1 if sharded else world)
        lde_gbs = stats["lde_algorithmic_bytes"] / (stats["lde_kernel_ms"] * 1e-3) / 1e9 if stats["lde_kernel_ms"] > 0 else 0.0
        # HBM bytes of the same LDE work from the PMC passes of tools/pmc_traffic.py (FETCH_SIZE x2 + WRITE_SIZE, calibrated
        # on a known copy as MI355X_MICROARCH.md prescribes), measured per column at the same log size and scaled to this
        # prove's column count; null when the committed measurement is for another size
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "fft_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("algorithmic_bytes_per_column") == float(16 << args.log_rows):
                    traffic = tj["hbm_bytes_per_column"] * (stats["lde_algorithmic_bytes"] / float(16 << args.log_rows))
            except Exception:
                traffic = None
        out = {
            "metric": "RISC-V cycles proved/sec at log_n_rows=%d; achieved HBM GB/s on Circle-FFT" % args.log_rows,
            "value": n_cycles / elapsed,
            "unit": "cycles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "u32 (M31)",
            "data": "synthetic",
            "config": {"workload": "synthetic 2^%d-row trace, full prove (LDE+quotient+FRI+Merkle), %d preprocessed + %d main + %d interaction columns, blowup 2, %d queries, pow_bits %d"
                       % (args.log_rows, args.n_pre, args.n_main, args.n_inter, cfg.n_queries, cfg.pow_bits),
                       "log_n_rows": args.log_rows, "parallelism": ("1 proof, columns sharded over %d GPUs" % world) if sharded else ("1 proof per GPU" if world > 1 else "1 GPU"),
                       "proof_words": int(len(words))},
            "roofline": {"bound": "hbm", "achieved": lde_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lde_gbs / HBM_PEAK_GBS,
                         "frac_of_measured_copy": lde_gbs / 6290.0,   # MI355X_MICROARCH.md: 6.29 TB/s measured copy bandwidth
                         "traffic": traffic, "kernel": "nx::fft13_kernel<INV, FIRST, 1, 4, K> + nx::lde_mid_kernel<4, K> (Circle iFFT+FFT = LDE, all passes of one prove)",
                         "algorithmic_bytes": stats["lde_algorithmic_bytes"], "kernel_ms": stats["lde_kernel_ms"]},
            # the FFT is VALU-issue bound on gfx950, not HBM bound: butterflies of one prove's iFFT + LDE work
            # (n/2 * N per iFFT, n * N per 2x LDE: the trivial top layer is not computed) against the measured chip ceiling of the
            # butterfly instruction sequence itself (tools/ubench/bfly_rates.hip, variant B: 4.73e12 butterflies/s)
            "roofline_valu": (lambda nb: {"bound": "valu", "achieved": nb / (stats["lde_kernel_ms"] * 1e-3) / 1e12, "peak": 4.73, "unit": "T butterflies/s",
                                          "frac": nb / (stats["lde_kernel_ms"] * 1e-3) / 4.73e12, "butterflies": nb})(
                (args.n_pre + args.n_main + args.n_inter) * 1.5 * args.log_rows * (1 << args.log_rows)
                + 4 * ((args.log_rows + 1) / 2.0 + (args.log_rows + 1)) * (2 << args.log_rows)),
            "stages_ms": {k: round(stats[k], 3) for k in ("trace_gen", "commit", "composition", "oods", "quotients", "fri", "pow", "decommit", "total")},
            "merkle": {"kernel_ms": stats["merkle_kernel_ms"], "algorithmic_bytes": stats["merkle_algorithmic_bytes"],
                       "achieved_GBs": stats["merkle_algorithmic_bytes"] / (stats["merkle_kernel_ms"] * 1e-3) / 1e9 if stats["merkle_kernel_ms"] > 0 else 0.0},
        }
        if not args.no_cpu_baseline and world == 1:
            import oracle_lib as O   # checker / baseline only — never part of the measured GPU path
            O.build_oracle()
            cores = os.cpu_count() or 1
            ccomps = [(args.cpu_log_rows, args.n_pre, args.n_main, args.n_inter)]
            secs = O.time_prove_synth(ccomps, O.default_cfg(pow_bits=args.pow_bits), seed=7, threads=cores)
            out["cpu_baseline"] = {"value": (1 << args.cpu_log_rows) / secs, "unit": "cycles/s", "cores": cores, "kind": "port",
                                   "sample": "one full prove of the same machine at 2^%d rows (%.1f s) by the C++ oracle, %d threads; NOT Stwo SimdBackend"
                                   % (args.cpu_log_rows, secs, cores)}
        print(json.dumps(out), flush=True)
    be.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
