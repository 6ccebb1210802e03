#!/bin/bash
# Round-4 evidence run on the GPU box: everything lands under gpurun_out/r04/ (copy what is to be judged into profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c 1-600
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt_bench -o kt -- python $R/bench.py --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 5 > /dev/null 2>&1)
python tools/rocprof_summary.py $O/kt_bench/kt_results.db $O/bench_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 5"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt_v1 -o kt -- python $R/bench.py --lcd 2 --n-logup 250 --extra-comps 8 --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 2 > /dev/null 2>&1)
python tools/rocprof_summary.py $O/kt_v1/kt_results.db $O/v1_shaped_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --lcd 2 --n-logup 250 --extra-comps 8 --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 2"
rm -rf $O/kt_bench $O/kt_v1
timeout 400 python tools/pmc_traffic.py --out $O/fft_traffic.json > /dev/null 2>&1
timeout 200 bash tools/r04_fft_time_ablation.sh $O/fft_time_ablation.jsonl > /dev/null 2>&1
for n in 16 18 20 22 24; do st=20; [ $n -ge 22 ] && st=5; timeout 300 python bench.py --log-rows $n --no-cpu-baseline --no-v1-shaped --no-host-trace --steps $st 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print(json.dumps({'log_rows': r['config']['log_n_rows'], 'ms_per_step': round(r['ms_per_step'], 3), 'cycles_per_s': r['value'], 'lde_ms': round(r['roofline']['kernel_ms'], 3), 'lde_alg_GBs': round(r['roofline']['achieved'], 1), 'stages_ms': r['stages_ms']}))" >> $O/bench_sizes.jsonl; done
cat $O/bench_sizes.jsonl | cut -c 1-200
timeout 300 python tools/keccak_shaped.py --steps 5 > $O/keccak_shaped.json 2>/dev/null
ls -la $O
