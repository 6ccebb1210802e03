#!/bin/bash
# A/B of "logup.staged"  on one box: nx_logup_cols' reads requested per group of fractions up front (1, the default)
# against the read-where-used kernel (0): parity of both against the oracle, then the bench's interaction stage, headline and v1-shaped.
# usage: bash tools/r05_logup_staged_ab.sh [out.txt]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=${1:-gpurun_out/r05/logup_staged_ab.txt}; mkdir -p "$(dirname "$out")"; : > "$out"
timeout 100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_machine.py -m gpu -q -x -p no:cacheprovider -k "logup_cols or logup_pipeline or logup_forms or machine_prove_equals" 2>&1 | tail -3 | tee -a "$out"
for rep in 1 2; do for v in 0 1; do
  NX_LOGUP_STAGED=$v timeout 100 python bench.py --steps 5 --no-cpu-baseline --no-host-trace 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); v1 = r['config_v1_shaped']
print('staged=$v headline %.2f ms (interaction %.3f)  v1-shaped %.2f ms (interaction %.3f)' % (r['ms_per_step'], r['stages_ms']['interaction'], v1['ms_per_step'], v1['stages_ms']['interaction']))" | tee -a "$out"
done; done
