cd "${GRAFT_REPO_ROOT:-/root/repo}"; export FFT_TUNE_MERKLE=0 FFT_TUNE_ROUNDS=8
timeout 300 python tools/fft_tune.py 22 128 3 fft.tile=0 fft.tile=1 fft.pipe=1 fft.pipe=1,fft.batch_cols=4
timeout 300 python tools/fft_tune.py 20 347 3 fft.tile=0 fft.tile=1 fft.pipe=1 fft.pipe=1,fft.batch_cols=4
timeout 300 python tools/fft_tune.py 18 438 3 fft.tile=0 fft.tile=1 fft.pipe=1
