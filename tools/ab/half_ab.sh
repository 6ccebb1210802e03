#!/bin/bash
# half-domain composition (NX_AIR_HALF_DOMAIN=1, default) vs the whole committed domain (0): parity tests, then headline and v1-shaped interleaved
timeout 1200 python -m pytest tests/test_gpu_machine.py tests/test_gpu_parity.py -m gpu -x -q -k "machine or session or prove or degree or air" 2>&1 | grep -E "passed|failed|Error" | tail -3
mkdir -p gpurun_out/ab
for round in 1 2; do
  for v in 0 1; do
    NX_AIR_HALF_DOMAIN=$v timeout 300 python bench.py --no-cpu-baseline --no-v1-shaped --steps 8 2>&1 | tail -1 > gpurun_out/ab/h_$v$round.json
    NX_AIR_HALF_DOMAIN=$v timeout 300 python bench.py --lcd 2 --n-logup 250 --extra-comps 8 --no-cpu-baseline --no-v1-shaped --steps 2 2>&1 | tail -1 > gpurun_out/ab/hv1_$v$round.json
    python - <<PY
import json
d=json.load(open("gpurun_out/ab/h_$v$round.json")); w=json.load(open("gpurun_out/ab/hv1_$v$round.json"))
print("half=$v", $round, "headline ms", round(d["ms_per_step"],3), "composition", d["stages_ms"]["composition"], "| v1-shaped ms", round(w["ms_per_step"],2), "composition", w["stages_ms"]["composition"])
PY
  done
done
