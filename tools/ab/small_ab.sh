#!/bin/bash
# small sizes, two builds interleaved (old = NX_LIB)
for n in 16 18; do
for round in 1 2 3; do
  for lib in old new; do
    if [ $lib = old ]; then export NX_LIB=$PWD/nexus-zkvm_amd/libnexus_hip_old.so; else unset NX_LIB; fi
    timeout 300 python bench.py --log-rows $n --no-cpu-baseline --no-v1-shaped --steps 20 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', $n, $round, round(d['ms_per_step'],3), d['stages_ms'])"
  done
done
done
