#!/bin/bash
# interleaved A/B of two library builds on one box: headline bench, Merkle / LDE kernel ms and total
mkdir -p gpurun_out/ab
for round in 1 2 3; do
  for lib in old new; do
    if [ $lib = old ]; then export NX_LIB=$PWD/nexus-zkvm_amd/libnexus_hip_old.so; else unset NX_LIB; fi
    timeout 300 python bench.py --no-cpu-baseline --no-v1-shaped --steps 10 2>&1 | tail -1 > gpurun_out/ab/$lib$round.json
    python - <<PY
import json
d=json.load(open("gpurun_out/ab/$lib$round.json"))
print("$lib", $round, "ms", round(d["ms_per_step"],3), "merkle", round(d["merkle"]["kernel_ms"],3), "lde", round(d["roofline"]["kernel_ms"],3), d["stages_ms"]["commit"])
PY
  done
done
