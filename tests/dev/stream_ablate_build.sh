#!/bin/bash
# dev: one libgptqhip per compile-time ablation mask of decode_stream_kernel -> tests/dev/ablate/libgptqhip_abl<N>.so (git-ignored, ships with gpurun)
set -e
cd "$(dirname "$0")/../../gptqmodel_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tests/dev/ablate
OBJS=$(ls *.o | grep -v gptqhip_stream.o)
for N in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -fno-gpu-rdc -DGPTQHIP_STREAM_ABLATE=$N -c gptqhip_stream.hip -o ../../tests/dev/ablate/gptqhip_stream_abl$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../tests/dev/ablate/gptqhip_stream_abl$N.o -o ../../tests/dev/ablate/libgptqhip_abl$N.so
done
echo built "$@"
