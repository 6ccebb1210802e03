# Time decomposition of the LDE kernels at 2^22 rows x 128 columns: the shipped build against two ablation builds made from
# tools/ab/patches/fft13_ablation.patch (sh tools/build_variant_lib.sh abl2 -DNX_FFT_ABL=2 / abl3 -DNX_FFT_ABL=3 on a patched copy of csrc):
#   abl3: neither LDS round trips nor butterflies (staging through LDS, global traffic, barriers only): the memory-system floor of the three launches
#   abl2: no butterfly arithmetic (LDS round trips, global traffic, barriers stay)
# Wrong results by design - timing only.  usage: bash tools/r04_fft_time_ablation.sh out.jsonl
out=${1:-gpurun_out/r04_fft_time_ablation.jsonl}
: > "$out"
for rep in 1 2; do for v in default abl2 abl3; do
  lib=""; [ $v != default ] && lib=$PWD/nexus-zkvm_amd/libnexus_hip_$v.so
  NX_LIB=$lib FFT_TUNE_MERKLE=0 python tools/fft_tune.py 22 128 4 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); r['variant'] = '$v'; r['rep'] = $rep; print(json.dumps(r))" >> "$out"
done; done
cat "$out"
