#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_machine.py tests/test_gpu_parity.py -m gpu -x -q -k "half or bit_exact or session_logup or degree or quotients" 2>&1 | grep -E "passed|failed|Abort|rror" | tail -2
for v in 0 1 0 1; do
  echo -n "half=$v: "; NX_AIR_HALF_DOMAIN=$v timeout 300 python bench.py --no-cpu-baseline --no-v1-shaped --steps 8 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['ms_per_step'], 3), d['stages_ms']['composition'])"
done
for v in 0 1 0 1; do
  echo -n "keccak half=$v: "; NX_AIR_HALF_DOMAIN=$v timeout 300 python tools/keccak_shaped.py --steps 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_prove'], d['stages_ms']['composition'])"
done
