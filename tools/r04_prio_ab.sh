# s_setprio in the LDE kernels (tools/ab/patches/fft13_prio.patch; sh tools/build_variant_lib.sh prio1 -DNX_FFT_PRIO=1 / prio2 -DNX_FFT_PRIO=2):
#   prio1: waves in a memory phase (tile requests, staging, stores) issue ahead of waves in butterfly rounds;  prio2: the reverse.
# Same results (the parity suite is not needed: no arithmetic changes); timing of 128 columns at 2^22 rows, then the whole prove.
# usage: bash tools/r04_prio_ab.sh out.jsonl
out=${1:-gpurun_out/r04_prio_ab.jsonl}
: > "$out"
for rep in 1 2; do for v in default prio1 prio2; do
  lib=""; [ $v != default ] && lib=$PWD/nexus-zkvm_amd/libnexus_hip_$v.so
  NX_LIB=$lib FFT_TUNE_MERKLE=0 python tools/fft_tune.py 22 128 4 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); r['variant'] = '$v'; r['rep'] = $rep; print(json.dumps(r))" >> "$out"
done; done
for v in default prio1 prio2; do
  lib=""; [ $v != default ] && lib=$PWD/nexus-zkvm_amd/libnexus_hip_$v.so
  NX_LIB=$lib python bench.py --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 8 --warmup 2 | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'variant': '$v', 'ms_per_prove': round(r['ms_per_step'], 3), 'commit_ms': r['stages_ms']['commit'], 'lde_kernel_ms': round(r['roofline']['kernel_ms'], 3)}))" >> "$out"
done
cat "$out"
