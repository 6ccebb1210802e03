# replicated whole-prove A/B of the shipped build against prio1 (see r04_prio_ab.sh).  usage: bash tools/r04_prio_ab2.sh out.jsonl
out=${1:-gpurun_out/r04_prio_ab2.jsonl}
: > "$out"
for rep in 1 2 3 4 5; do for v in default prio1; do
  lib=""; [ $v != default ] && lib=$PWD/nexus-zkvm_amd/libnexus_hip_$v.so
  NX_LIB=$lib python bench.py --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 10 --warmup 2 | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'variant': '$v', 'rep': $rep, 'ms_per_prove': round(r['ms_per_step'], 3), 'commit_ms': r['stages_ms']['commit'], 'lde_kernel_ms': round(r['roofline']['kernel_ms'], 3)}))" >> "$out"
done; done
cat "$out"
