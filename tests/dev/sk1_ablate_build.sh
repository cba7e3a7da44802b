#!/bin/bash
# dev: one libgptqhip per compile-time ablation mask of skinny1_kernel (GPTQHIP_SK1_ABLATE) -> tests/dev/ablate/libgptqhip_sk1abl<N>.so (git-ignored, ships
# with gpurun).  Masks: 1 no small-operand loads, 2 no dequant / MFMA, 4 no reduction / epilogue, 8 no weight loads.
set -e
cd "$(dirname "$0")/../../gptqmodel_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tests/dev/ablate
OBJS=$(ls *.o | grep -v gptqhip_skinny.o)
build() {
  N=$1
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -fno-gpu-rdc -DGPTQHIP_SK1_ABLATE=$N -c gptqhip_skinny.hip -o ../../tests/dev/ablate/gptqhip_skinny_sk1abl$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../tests/dev/ablate/gptqhip_skinny_sk1abl$N.o -o ../../tests/dev/ablate/libgptqhip_sk1abl$N.so
}
for N in "$@"; do build $N & done
wait
echo built "$@"
