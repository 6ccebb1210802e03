#!/bin/sh
# A/B build of the library with extra compiler flags -> nexus-zkvm_amd/libnexus_hip_<tag>.so (load it with NX_LIB=<path>, tools only).
# usage: sh tools/build_variant_lib.sh mw8 -DNX_FFT_MINWAVES14=8
set -e
tag=$1; shift
cd "$(dirname "$0")/../nexus-zkvm_amd/csrc"
T=$(mktemp -d)
for f in ctx fft fft13 merkle pcs air constraints air_jit logup backend_ops prover machine serde comm_rccl comm_local; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -Wno-unused-result -c $f.hip -o $T/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnexus_hip_$tag.so $T/*.o -L/opt/rocm/lib -lhiprtc -ldl
rm -rf $T
echo built nexus-zkvm_amd/libnexus_hip_$tag.so
