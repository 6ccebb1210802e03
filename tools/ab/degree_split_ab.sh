#!/bin/bash
# degree-split: parity tests, then the v1-shaped workload with and without it
mkdir -p gpurun_out/ds
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_machine.py -m gpu -x -q -k "subset or degree_split or session_logup or machine_prove_bit_exact or v1_shaped or air_jit" 2>&1 | tail -8
for s in 1 0; do
  NX_AIR_DEGREE_SPLIT=$s timeout 600 python bench.py --lcd 2 --n-logup 250 --extra-comps 8 --no-cpu-baseline --no-v1-shaped --steps 3 2>&1 | tail -1 > gpurun_out/ds/v1_split$s.json
  python - <<PY
import json
d=json.load(open("gpurun_out/ds/v1_split$s.json"))
print("split=$s", round(d["ms_per_step"],2), d["stages_ms"])
PY
done
