#!/bin/bash
# DEEP quotients from the coefficient columns (NX_QUOTIENTS_COEFFS=1, default) vs row-wise over the extensions (0): parity tests, then
# the headline and the v1-shaped workload interleaved on one box
timeout 900 python -m pytest tests/test_gpu_machine.py tests/test_gpu_parity.py -m gpu -x -q -k "quotient or machine_prove_bit_exact or protocol_switch or prove" 2>&1 | grep -E "passed|failed" | tail -2
mkdir -p gpurun_out/ab
for round in 1 2; do
  for v in 0 1; do
    NX_QUOTIENTS_COEFFS=$v timeout 300 python bench.py --no-cpu-baseline --no-v1-shaped --steps 8 2>&1 | tail -1 > gpurun_out/ab/q_$v$round.json
    NX_QUOTIENTS_COEFFS=$v timeout 300 python bench.py --lcd 2 --n-logup 250 --extra-comps 8 --no-cpu-baseline --no-v1-shaped --steps 2 2>&1 | tail -1 > gpurun_out/ab/qv1_$v$round.json
    python - <<PY
import json
d=json.load(open("gpurun_out/ab/q_$v$round.json")); w=json.load(open("gpurun_out/ab/qv1_$v$round.json"))
print("coeffs=$v", $round, "headline ms", round(d["ms_per_step"],3), "quotients", d["stages_ms"]["quotients"], "| v1-shaped ms", round(w["ms_per_step"],2), "quotients", w["stages_ms"]["quotients"])
PY
  done
done
