#!/bin/bash
# dev: build timing-ablation variants of the decode kernel (results are WRONG by construction; timing only)
cd /root/repo/gptqmodel_amd/csrc
for a in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc -DGPTQHIP_ABLATE=$a -c gptqhip_skinny.hip -o /tmp/skinny_abl$a.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gptqhip_abi.o /tmp/skinny_abl$a.o gptqhip_tiled.o gptqhip_tiled_f32.o gptqhip_tiled8.o gptqhip_aux.o gptqhip_comm.o -o ../../tests/dev/ablate/libgptqhip_abl$a.so &
done
wait
ls -la ../../tests/dev/ablate/
