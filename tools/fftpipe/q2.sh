cd "${GRAFT_REPO_ROOT:-/root/repo}"; export FFT_TUNE_MERKLE=0
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined_lde_item_loop" 2>&1 | tail -3
timeout 200 python tools/fft_tune.py 22 128 4 fft.tile=0 fft.tile=1 fft.tile=1,fft.batch_cols=4 fft.tile=1,fft.streams=1 fft.tile=1,fft.streams=3 fft.tile=1,fft.batch_cols=1,fft.streams=4 fft.tile=0
timeout 200 python tools/fft_tune.py 20 347 4 fft.tile=0 fft.tile=1
