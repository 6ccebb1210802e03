# Whole-prove A/B: host set-up overlapped with queued GPU work (this tree) against the previous order (libnexus_hip_base.so built from the
# commit before).  Alternating pairs on one box.  usage: bash tools/r04_setup_overlap_ab.sh out.jsonl
out=${1:-gpurun_out/r04_setup_overlap_ab.jsonl}
: > "$out"
for rep in 1 2 3 4 5 6; do for v in base new; do
  lib=""; [ $v = base ] && lib=$PWD/nexus-zkvm_amd/libnexus_hip_base.so
  NX_LIB=$lib python bench.py --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 12 --warmup 3 | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'variant': '$v', 'rep': $rep, 'ms_per_prove': round(r['ms_per_step'], 3)}))" >> "$out"
done; done
cat "$out"
