#!/bin/sh
# Ablation builds of the pipelined FFT kernels (timing only): libnexus_hip_abl{1,2,3}.so next to the product library.
# 1 = no butterfly arithmetic, 2 = no global traffic (no DMA, no stores), 3 = both (LDS round trips + barriers only).
set -e
cd "$(dirname "$0")/../../nexus-zkvm_amd/csrc"
make -j8 -s
for a in 1 2 3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DNX_PIPE_ABL=$a -Wno-unused-result -Wno-unused-function -c fft_pipe.hip -o /tmp/fft_pipe_abl$a.o &
done
wait
OBJS=$(ls *.o | grep -v "fft_pipe")
for a in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnexus_hip_abl$a.so $OBJS /tmp/fft_pipe_abl$a.o -L/opt/rocm/lib -lhiprtc
done
ls -la ../libnexus_hip_abl*.so
