#!/bin/bash
# Round-5 evidence run on the GPU box: everything lands under gpurun_out/r05/ (copy what is to be judged into profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
# 1. the driver's line (N = 1): headline + v1-shaped + host trace + preprocessed reuse + cpu_baseline
timeout 900 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c 1-400
# 2. rocprofv3 kernel traces of the same command (headline) and of the v1-shaped statement
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt_bench -o kt -- python $R/bench.py --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 5 > /dev/null 2>&1)
python tools/rocprof_summary.py $O/kt_bench/kt_results.db $O/bench_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 5"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt_v1 -o kt -- python $R/bench.py --lcd 2 --n-logup 250 --extra-comps 8 --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 2 > /dev/null 2>&1)
python tools/rocprof_summary.py $O/kt_v1/kt_results.db $O/v1_shaped_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --lcd 2 --n-logup 250 --extra-comps 8 --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 2"
rm -rf $O/kt_bench $O/kt_v1
# 3. HBM-side traffic of the Circle-FFT LDE (separate --pmc passes)
timeout 400 python tools/pmc_traffic.py --out $O/fft_traffic.json > /dev/null 2>&1
# 4. the headline statement at other sizes
for n in 16 18 20 22 24; do st=20; [ $n -ge 22 ] && st=5; timeout 300 python bench.py --log-rows $n --no-cpu-baseline --no-v1-shaped --no-host-trace --steps $st 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print(json.dumps({'log_rows': r['config']['log_n_rows'], 'ms_per_step': round(r['ms_per_step'], 3), 'cycles_per_s': r['value'], 'lde_ms': round(r['roofline']['kernel_ms'], 3), 'lde_alg_GBs': round(r['roofline']['achieved'], 1), 'stages_ms': r['stages_ms']}))" >> $O/bench_sizes.jsonl; done
# 5. config #5 in the reference's logup forms (pairs + tables over preprocessed columns) and in round 4's form; the first prove of a fresh
#    process with an empty and with a filled kernel cache directory (nx_air_cache_dir)
rm -rf /tmp/nxair; 
NX_AIR_CACHE_DIR=/tmp/nxair timeout 300 python tools/keccak_shaped.py --steps 5 > $O/keccak_shaped_pairs.json 2>/dev/null
NX_AIR_CACHE_DIR=/tmp/nxair timeout 300 python tools/keccak_shaped.py --steps 1 > $O/keccak_shaped_pairs_warm_cache.json 2>/dev/null
timeout 300 python tools/keccak_shaped.py --steps 5 --single > $O/keccak_shaped_single.json 2>/dev/null
# 6. config #2: 347 columns x 2^20 rows, LDE + Merkle, uniform and byte-limb values
timeout 200 python tools/fft_tune.py 20 347 5 > $O/config2_uniform.jsonl 2>/dev/null
FFT_TUNE_BYTE_LIMBS=1 timeout 200 python tools/fft_tune.py 20 347 5 > $O/config2_byte_limbs.jsonl 2>/dev/null
# 7. ONE proof on 8 thread-ranks of this GPU: the collectives a rank enters (the library's own counters) and the sharding overhead proxy
timeout 600 python tools/thread_ranks_bench.py 22 8 > $O/thread_ranks_proxy.json 2>/dev/null
ls -la $O
