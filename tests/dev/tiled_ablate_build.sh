#!/bin/bash
# dev: timing-ablation builds of the prefill kernel (WRONG results by construction): $1 = tag, $2.. = -D flags
cd /root/repo/gptqmodel_amd/csrc
mkdir -p ../../tests/dev/ablate
tag=$1; shift
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc $*"
TUS="gptqhip_tiled gptqhip_tiled_f32 gptqhip_tiled8 gptqhip_tiled_r32 gptqhip_tiled_r48 gptqhip_tiled_r80 gptqhip_tiled_r96 gptqhip_tiled_r112 gptqhip_tiled_n128_r32 gptqhip_tiled_n128_r48 gptqhip_tiled_n128_r64 gptqhip_tiled_n128_r80 gptqhip_tiled_n128_r96 gptqhip_tiled_n128_r112 gptqhip_tiled_n128_r128"
for f in $TUS; do
  /opt/rocm/bin/hipcc $FL -c $f.hip -o /tmp/${f}_$tag.o &
done
wait
OBJS=""; for f in $TUS; do OBJS="$OBJS /tmp/${f}_$tag.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gptqhip_abi.o gptqhip_skinny.o gptqhip_aux.o gptqhip_comm.o $OBJS -o ../../tests/dev/ablate/libgptqhip_$tag.so
ls -la ../../tests/dev/ablate/libgptqhip_$tag.so
