#!/bin/bash
# dev: build libgptqhip with per-block phase stamps in the stripe kernel (tests/dev/stripe_stamps.py reads them)
cd /root/repo/gptqmodel_amd/csrc
mkdir -p ../../tests/dev/ablate
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc -DGPTQHIP_STRIPE_STAMPS $STRIPE_EXTRA"
for f in gptqhip_stripe gptqhip_stripe_a1s0 gptqhip_stripe_a0s1 gptqhip_stripe_a1s1; do
  /opt/rocm/bin/hipcc $FL -c $f.hip -o /tmp/${f}_st.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gptqhip_abi.o gptqhip_skinny.o gptqhip_tiled.o gptqhip_tiled_f32.o gptqhip_tiled8.o gptqhip_aux.o gptqhip_comm.o \
  /tmp/gptqhip_stripe_st.o /tmp/gptqhip_stripe_a1s0_st.o /tmp/gptqhip_stripe_a0s1_st.o /tmp/gptqhip_stripe_a1s1_st.o -o ../../tests/dev/ablate/libgptqhip_stamps${STRIPE_TAG}.so
ls -la ../../tests/dev/ablate/
