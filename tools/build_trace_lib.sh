#!/bin/sh
# A/B build of the library with the per-block phase timers of the fft13 kernels -> nexus-zkvm_amd/libnexus_hip_trace.so
# The instrumentation is NOT in the product sources (VERDICT r3 hygiene): tools/ab/patches/fft13_trace.patch adds it to a scratch copy.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
T=$(mktemp -d)
cp -r "$ROOT/nexus-zkvm_amd/csrc" "$T/csrc"; mkdir -p "$T/include" && cp "$ROOT/include/nexus_hip.h" "$T/include/"
(cd "$T" && mkdir -p nexus-zkvm_amd && mv csrc nexus-zkvm_amd/ && patch -p0 < "$ROOT/tools/ab/patches/fft13_trace.patch")
cd "$T/nexus-zkvm_amd/csrc"
for f in ctx fft fft13 merkle pcs air constraints air_jit logup backend_ops prover machine serde comm_rccl comm_local; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DNX_FFT_TRACE -Wno-unused-result -c $f.hip -o $f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/nexus-zkvm_amd/libnexus_hip_trace.so" *.o -L/opt/rocm/lib -lhiprtc -ldl
rm -rf "$T"
