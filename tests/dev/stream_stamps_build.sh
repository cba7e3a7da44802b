#!/bin/bash
# dev: libgptqhip with per-wave phase clocks in decode_stream_kernel -> tests/dev/ablate/libgptqhip_stamps.so (git-ignored, ships with gpurun)
set -e
cd "$(dirname "$0")/../../gptqmodel_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tests/dev/ablate
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc -DGPTQHIP_STREAM_STAMPS=1 -c gptqhip_stream.hip -o ../../tests/dev/ablate/gptqhip_stream_stamps.o
OBJS=$(ls *.o | grep -v gptqhip_stream.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../tests/dev/ablate/gptqhip_stream_stamps.o -o ../../tests/dev/ablate/libgptqhip_stamps.so
echo built
