#!/bin/bash
# Run ON THE GPU BOX: parity of decode form 5 (raw codes as fp16 denormals) + A/B against forms 3 / 4 / 0 on the Llama-3-8B chain.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_form5_ab.txt
python -m pytest $R/tests/test_gpu_decode_forms.py -x -q 2>&1 | tail -8 > $O
AB_FORMS=0,3,5,4,3,5 python $R/tests/dev/decode_ab.py fp16 2>&1 | grep -v amdgpu.ids >> $O
cat $O
