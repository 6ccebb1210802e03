cd "${GRAFT_REPO_ROOT:-/root/repo}"; export FFT_TUNE_MERKLE=0; R=$PWD
for l in "" _hoist; do echo "lib$l"; NX_LIB=$R/nexus-zkvm_amd/libnexus_hip$l.so timeout 120 python tools/fft_tune.py 22 64 4 fft.pipe=1 fft.pipe=1,fft.batch_cols=4 fft.pipe=0; done
