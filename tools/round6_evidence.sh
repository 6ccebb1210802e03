#!/bin/bash
# Round-6 evidence run on the GPU box: everything lands under gpurun_out/r06/ev/ (copy what is to be judged into profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=${EV_OUT:-gpurun_out/r06/ev}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
# 1. the driver's line (N = 1): headline + v1-shaped (+ the reference's tuple widths) + host trace + preprocessed reuse + cpu_baseline
timeout 1200 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c 1-300
# 2. rocprofv3 kernel traces of the same command (headline) and of the v1-shaped statement; the kernel sequence / GPU idle gaps of one headline prove
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_bench -o kt -- python bench.py --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 5 > /dev/null 2>&1
python tools/rocprof_summary.py $O/kt_bench/kt_results.db $O/bench_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 5"
# (kernel sequences come from step 2b: bench.py ends with a statistics prove, whose stages end in synchronisations — not what a timed prove does)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_v1 -o kt -- python bench.py --lcd 2 --n-logup 250 --extra-comps 8 --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 2 > /dev/null 2>&1
python tools/rocprof_summary.py $O/kt_v1/kt_results.db $O/v1_shaped_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --lcd 2 --n-logup 250 --extra-comps 8 --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 2"
rm -rf $O/kt_bench $O/kt_v1
# 2b. where the GPU idles inside a TIMED prove: kernel sequences of plain proves (no statistics call) of the headline, the v1 shape and config #5 at the reference's tuple widths
for W in headline v1 keccakw; do
  rm -rf /tmp/kt_$W
  timeout 600 rocprofv3 --kernel-trace -d /tmp/kt_$W -o kt -- python tools/prove_loop.py $W --steps 2 > /dev/null 2>&1
  python tools/kernel_sequence.py $(find /tmp/kt_$W -name '*_results.db' | head -1) $O/seq_nostats_$W.txt
  timeout 300 python tools/prove_loop.py $W --steps 5 >> $O/prove_loop.jsonl 2>/dev/null
  rm -rf /tmp/kt_$W
done
NX_HOST_PROF=1 timeout 300 python tools/prove_loop.py keccakw --steps 1 > /dev/null 2> $O/host_prof_keccakw.txt
NX_HOST_PROF=1 timeout 300 python tools/prove_loop.py headline --steps 1 > /dev/null 2> $O/host_prof_headline.txt
timeout 100 tools/ubench/mall_bw > $O/mall_bw.jsonl 2>/dev/null
# 3. HBM-side traffic of the Circle-FFT LDE (separate --pmc passes)
timeout 400 python tools/pmc_traffic.py --out $O/fft_traffic.json > /dev/null 2>&1
# 4. the headline statement at other sizes
for n in 16 18 20 22 24; do st=20; [ $n -ge 22 ] && st=5; timeout 300 python bench.py --log-rows $n --no-cpu-baseline --no-v1-shaped --no-host-trace --steps $st 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print(json.dumps({'log_rows': r['config']['log_n_rows'], 'ms_per_step': round(r['ms_per_step'], 3), 'cycles_per_s': r['value'], 'lde_ms': round(r['roofline']['kernel_ms'], 3), 'lde_alg_GBs': round(r['roofline']['achieved'], 1), 'stages_ms': r['stages_ms']}))" >> $O/bench_sizes.jsonl; done
# 5. config #5: the reference's logup forms with one- / two-column tuples (round 5's statement), at the reference's tuple WIDTHS (round 6), the
#    first prove of a fresh process cold (kernels compiled in helper processes) and with a filled kernel cache directory
rm -rf /tmp/nxair ~/.cache/comgr
timeout 300 python tools/keccak_shaped.py --steps 5 > $O/keccak_shaped_pairs.json 2>/dev/null
rm -rf ~/.cache/comgr
NX_AIR_CACHE_DIR=/tmp/nxair timeout 300 python tools/keccak_shaped.py --steps 5 --tuples > $O/keccak_shaped_tuples.json 2>/dev/null
NX_AIR_CACHE_DIR=/tmp/nxair timeout 300 python tools/keccak_shaped.py --steps 1 --tuples > $O/keccak_shaped_tuples_warm_cache.json 2>/dev/null
rm -rf ~/.cache/comgr
NX_AIR_COMPILE_PROCS=1 timeout 300 python tools/keccak_shaped.py --steps 1 --tuples > $O/keccak_shaped_tuples_one_process.json 2>/dev/null
# 6. config #2: 347 columns x 2^20 rows, LDE + Merkle, uniform and byte-limb values
timeout 200 python tools/fft_tune.py 20 347 5 > $O/config2_uniform.jsonl 2>/dev/null
FFT_TUNE_BYTE_LIMBS=1 timeout 200 python tools/fft_tune.py 20 347 5 > $O/config2_byte_limbs.jsonl 2>/dev/null
# 7. ONE proof on 8 thread-ranks of this GPU: the collectives a rank enters (the library's own counters) and the sharding overhead proxy;
#    the v1-shaped statement with and without the quarter domain inside the sharded proof
timeout 600 python tools/thread_ranks_bench.py 22 8 > $O/thread_ranks_proxy.json 2>/dev/null
for o in "-" "air.quarter_domain=0"; do timeout 600 python tools/thread_ranks_bench.py 20 8 $o v1 2>/dev/null | tail -1 >> $O/thread_ranks_v1_quarter.jsonl; done
ls -la $O
