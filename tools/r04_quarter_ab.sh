# A/B of "air.quarter_domain"  on one box: the v1-shaped workload of bench.py with the option off and on, twice each.
# usage: bash tools/r04_quarter_ab.sh out.jsonl
out=${1:-gpurun_out/r04_quarter_ab.jsonl}
: > "$out"
for rep in 1 2; do for q in 0 1; do
  NX_AIR_QUARTER_DOMAIN=$q python bench.py --no-cpu-baseline --steps 3 --warmup 1 | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); v = r['config_v1_shaped']
print(json.dumps({'air.quarter_domain': $q, 'rep': $rep, 'headline_ms': r['ms_per_step'], 'v1_ms': v['ms_per_step'], 'v1_cycles_per_s': v['value'], 'v1_stages_ms': v['stages_ms'], 'v1_lde_kernel_ms': v['roofline']['kernel_ms']}))" >> "$out"
done; done
cat "$out"
