#!/bin/bash
# Where do the pipelined kernels' cycles go?  ablations, per-kernel durations, SQ counters.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/fftpipe2; mkdir -p $O
export FFT_TUNE_MERKLE=0 TMPDIR=/tmp
R=$PWD
for l in "" _abl1 _abl2 _abl3; do
  echo "lib$l" >> $O/abl.log
  NX_LIB=$R/nexus-zkvm_amd/libnexus_hip$l.so timeout 120 python tools/fft_tune.py 22 64 3 fft.pipe=1 fft.pipe=1,fft.batch_cols=8,fft.streams=1 fft.pipe=1,fft.pipe_blocks_per_cu=1,fft.streams=1,fft.batch_cols=8 >> $O/abl.log 2>&1
done
cat $O/abl.log
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/kt_pipe -o kt -- python $R/tools/fft_tune.py 22 64 2 fft.pipe=1,fft.streams=1 > /dev/null 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/kt_old -o kt -- python $R/tools/fft_tune.py 22 64 2 fft.pipe=0,fft.streams=1 > /dev/null 2>&1)
for d in kt_pipe kt_old; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d $f"; head -8 "$f" | cut -c 1-200; done
timeout 400 python tools/pmc_sq.py --cols 32 --opts fft.pipe=1 --passes 0,1 --out $O/sq_pipe.json > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/fftpipe2/sq_pipe.json"))
for k,v in d["kernels"].items():
    if "pipe" in k or "_errors" in k:
        print(k[:60], {a: v.get(a) for a in ("frac_of_wave_cycles","SQ_INSTS_VALU","SQ_INSTS_LDS","SQ_INSTS_SALU","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_WAVES","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES")} if isinstance(v, dict) else v)
PY
