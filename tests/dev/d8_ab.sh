#!/bin/bash
# Run ON THE GPU BOX: A/B of the 8-deep ring geometries of the preload decode kernels (dev switches GPTQHIP_SK1_D8 / GPTQHIP_SK1P_D8), one process
# per variant (the switches are read once), two rounds so that drift shows.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_d8_ab_form5.txt
: > $O
for round in 1 2; do
  for v in "base" "GPTQHIP_SK1_D8=1" "GPTQHIP_SK1_D8=2" "GPTQHIP_SK1P_D8=1" "GPTQHIP_SK1_D8=2 GPTQHIP_SK1P_D8=1" "GPTQHIP_NO_PERSIST=1 GPTQHIP_SK1_D8=2"; do
    echo "## $v" >> $O
    if [ "$v" = "base" ]; then AB_FORMS=0,5,5 python $R/tests/dev/decode_ab.py fp16 >> $O 2>&1
    else env $v AB_FORMS=0,5,5 python $R/tests/dev/decode_ab.py fp16 >> $O 2>&1; fi
  done
done
GPTQHIP_SK1_D8=2 GPTQHIP_SK1P_D8=1 python -m pytest $R/tests/test_gpu_decode_forms.py -x -q 2>&1 | tail -5 >> $O
cat $O
