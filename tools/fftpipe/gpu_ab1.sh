#!/bin/bash
# First GPU contact of the pipelined FFT kernels: parity first, then an option sweep, then the bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/fftpipe1; mkdir -p $O
export FFT_TUNE_MERKLE=0
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined_lde_item_loop or context_options or fused_lde_matches" > $O/pytest_pipe.log 2>&1
tail -3 $O/pytest_pipe.log
if grep -q "failed\|error" $O/pytest_pipe.log; then echo PARITY_FAILED; tail -40 $O/pytest_pipe.log; fi
timeout 300 python tools/fft_tune.py 22 128 3 fft.pipe=0 fft.pipe=1 fft.pipe=1,fft.batch_cols=4 fft.pipe=1,fft.batch_cols=8 fft.pipe=1,fft.streams=1 fft.pipe=1,fft.batch_cols=4,fft.streams=1 \
   fft.pipe=1,fft.batch_cols=8,fft.streams=1 fft.pipe=1,fft.batch_cols=16,fft.streams=1 fft.pipe=1,fft.pipe_blocks_per_cu=1 fft.pipe=1,fft.pipe_blocks_per_cu=1,fft.batch_cols=4 \
   fft.pipe=1,fft.batch_cols=4,fft.streams=3 fft.pipe=1,fft.batch_cols=1,fft.streams=4 fft.pipe=0,fft.batch_cols=4,fft.streams=1 > $O/sweep22.jsonl 2>&1
cat $O/sweep22.jsonl
timeout 200 python tools/fft_tune.py 20 347 3 fft.pipe=0 fft.pipe=1 fft.pipe=1,fft.batch_cols=4 fft.pipe=1,fft.batch_cols=4,fft.streams=1 > $O/sweep20.jsonl 2>&1
cat $O/sweep20.jsonl
timeout 300 python bench.py --no-cpu-baseline --no-v1-shaped --steps 10 > $O/bench_pipe.json 2> $O/bench_pipe.err; tail -1 $O/bench_pipe.json | cut -c 1-1500
NX_FFT_PIPE=0 timeout 300 python bench.py --no-cpu-baseline --no-v1-shaped --steps 10 > $O/bench_old.json 2> $O/bench_old.err; tail -1 $O/bench_old.json | cut -c 1-600
