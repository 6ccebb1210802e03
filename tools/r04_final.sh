#!/bin/bash
# Round-4 closing run on the GPU box: the bench line as the driver runs it, and the rocprofv3 kernel summary of the same command.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04final; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 120 python -c "import nexus_zkvm_amd as nz; be = nz.HipBackend(0); be.precompute_twiddles(10); be.sync(); print('preflight ok')" 2>&1 | tail -1 || true
timeout 500 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c 1-600
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt_bench -o kt -- python $R/bench.py --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 5 > /dev/null 2> $R/$O/kt.err)
python tools/rocprof_summary.py $O/kt_bench/kt_results.db $O/bench_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 5"
python tools/trace_gaps.py $O/kt_bench/kt_results.db 25 2 > $O/trace_gaps.txt 2>&1
rm -rf $O/kt_bench
head -12 $O/bench_kernel_stats.txt | cut -c1-150; head -14 $O/trace_gaps.txt | cut -c1-150
