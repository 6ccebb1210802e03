# A/B of the 2^14-row FFT tiles (two passes for >= 2^23 points) against the round-3 plan (2^13-row tiles, three passes), same box.
# usage: bash tools/r04_tile14_ab.sh out.jsonl      (NX_LIB variant: nexus-zkvm_amd/libnexus_hip_mw8.so = -DNX_FFT_MINWAVES14=8, if built)
out=${1:-gpurun_out/r04_tile14_ab.jsonl}
: > "$out"
for lg in 23 24; do
  for mode in "0:" "1:" "1:$PWD/nexus-zkvm_amd/libnexus_hip_mw8.so"; do
    t=${mode%%:*}; lib=${mode#*:}
    if [ -n "$lib" ] && [ ! -f "$lib" ]; then continue; fi
    NX_FFT_TILE14=$t NX_LIB=$lib FFT_TUNE_MERKLE=0 python tools/fft_tune.py $lg 32 4 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); r['tile14'] = $t; r['lib'] = '$lib'.split('/')[-1] or 'default'; print(json.dumps(r))" >> "$out"
  done
done
for t in 0 1; do
  NX_FFT_TILE14=$t python bench.py --log-rows 24 --steps 2 --warmup 1 --no-cpu-baseline --no-v1-shaped --no-host-trace | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'tile14': $t, 'bench_log_rows': 24, 'ms_per_prove': r['ms_per_step'], 'cycles_per_s': r['value'], 'lde_kernel_ms': r['roofline']['kernel_ms'], 'lde_alg_GBs': r['roofline']['achieved'], 'frac': r['roofline']['frac'], 'stages_ms': r['stages_ms']}))" >> "$out"
done
cat "$out"
