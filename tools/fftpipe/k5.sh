cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/k5; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_gpu_machine.py -m gpu -x -q -k "config5_keccak_shaped_at_full_width" 2>&1 | tail -4
timeout 600 python tools/keccak_shaped.py --steps 3 > gpurun_out/k5/keccak_shaped.json 2> gpurun_out/k5/err.log; cat gpurun_out/k5/keccak_shaped.json; tail -3 gpurun_out/k5/err.log
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/k5/kt -o kt -- python $R/tools/keccak_shaped.py --steps 2 > /dev/null 2>&1)
python tools/rocprof_summary.py gpurun_out/k5/kt/kt_results.db gpurun_out/k5/keccak_shaped_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python tools/keccak_shaped.py --steps 2"; head -25 gpurun_out/k5/keccak_shaped_kernel_stats.txt
