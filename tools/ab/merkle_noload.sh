#!/bin/bash
# Merkle commit of 347 columns x 2^23 rows: the shipped build against the ablation build whose message words come from registers
# (libnexus_hip_abl.so: a scratch copy of csrc with tools/ab/patches/merkle_noload.patch applied, compiled with -DNX_MERKLE_ABL_NOLOAD;
#  wrong hashes by design, timing only — the ablation is not in the product sources)
for round in 1 2 3; do
  for lib in real noload; do
    if [ $lib = noload ]; then export NX_LIB=$PWD/nexus-zkvm_amd/libnexus_hip_abl.so; else unset NX_LIB; fi
    echo -n "$lib $round: "; timeout 300 python tools/fft_tune.py 22 347 3 2>/dev/null | tail -1 | cut -c 1-400
  done
done
