#!/bin/bash
# keccak-shaped workload, two builds interleaved (old = NX_LIB)
for round in 1 2 3; do
  for lib in old new; do
    if [ $lib = old ]; then export NX_LIB=$PWD/nexus-zkvm_amd/libnexus_hip_old.so; else unset NX_LIB; fi
    timeout 300 python tools/keccak_shaped.py --steps 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', $round, d['ms_per_prove'], d['stages_ms'])"
  done
done
