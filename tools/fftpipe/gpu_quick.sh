#!/bin/bash
# parity of the pipelined LDE + a short timing line (product lib and the three ablation builds)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/fftpipe_q; mkdir -p $O
export FFT_TUNE_MERKLE=0 TMPDIR=/tmp
R=$PWD
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined_lde_item_loop or fused_lde_matches" 2>&1 | tail -2
for l in "" _abl1 _abl2 _abl3; do
  echo "lib$l"
  NX_LIB=$R/nexus-zkvm_amd/libnexus_hip$l.so timeout 120 python tools/fft_tune.py 22 64 3 fft.pipe=1 fft.pipe=1,fft.batch_cols=4 $EXTRA
done
timeout 120 python tools/fft_tune.py 22 64 3 fft.pipe=0
