#!/bin/bash
# dev: timing-ablation builds of the prefill kernel (WRONG results by construction): $1 = tag, $2.. = -D flags
cd /root/repo/gptqmodel_amd/csrc
mkdir -p ../../tests/dev/ablate
tag=$1; shift
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc $*"
for f in gptqhip_tiled gptqhip_tiled_f32 gptqhip_tiled8; do
  /opt/rocm/bin/hipcc $FL -c $f.hip -o /tmp/${f}_$tag.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gptqhip_abi.o gptqhip_skinny.o gptqhip_aux.o gptqhip_comm.o gptqhip_stripe.o gptqhip_stripe_a1s0.o gptqhip_stripe_a0s1.o gptqhip_stripe_a1s1.o \
  /tmp/gptqhip_tiled_$tag.o /tmp/gptqhip_tiled_f32_$tag.o /tmp/gptqhip_tiled8_$tag.o -o ../../tests/dev/ablate/libgptqhip_$tag.so
ls -la ../../tests/dev/ablate/libgptqhip_$tag.so
