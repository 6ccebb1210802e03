#!/bin/sh
# A/B build of the library with -DNX_FFT_TRACE (per-block phase timers in the fft13 kernels) -> nexus-zkvm_amd/libnexus_hip_trace.so
set -e
cd "$(dirname "$0")/../nexus-zkvm_amd/csrc"
T=$(mktemp -d)
for f in ctx fft fft13 merkle pcs air constraints air_jit logup prover; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DNX_FFT_TRACE -Wno-unused-result -c $f.hip -o $T/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnexus_hip_trace.so $T/*.o -L/opt/rocm/lib -lhiprtc
rm -rf $T
