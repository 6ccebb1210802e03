#!/bin/bash
# logup parity tests, then interleaved A/B (old build via NX_LIB) of the interaction stage: headline and v1-shaped
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_machine.py -m gpu -x -q -k "logup or machine_prove_bit_exact or wide" 2>&1 | tail -4
mkdir -p gpurun_out/ab
for round in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export NX_LIB=$PWD/nexus-zkvm_amd/libnexus_hip_old.so; else unset NX_LIB; fi
    timeout 300 python bench.py --no-cpu-baseline --no-v1-shaped --steps 5 2>&1 | tail -1 > gpurun_out/ab/lg_$lib$round.json
    timeout 300 python bench.py --lcd 2 --n-logup 250 --extra-comps 8 --no-cpu-baseline --no-v1-shaped --steps 2 2>&1 | tail -1 > gpurun_out/ab/lgv1_$lib$round.json
    python - <<PY
import json
d=json.load(open("gpurun_out/ab/lg_$lib$round.json")); v=json.load(open("gpurun_out/ab/lgv1_$lib$round.json"))
print("$lib", $round, "headline ms", round(d["ms_per_step"],3), "interaction", d["stages_ms"]["interaction"], "| v1-shaped ms", round(v["ms_per_step"],2), "interaction", v["stages_ms"]["interaction"], "composition", v["stages_ms"]["composition"])
PY
  done
done
