#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the prefill kernel, fp16 vs bf16 (activations + scales), M = 8192 and 2048 on 4096^2 (VERDICT r5 item 4:
# is the bf16 K loop VALU-issue-bound or register-bound?).  One counter group per pass.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  for dt in fp16 bf16; do
    for m in 8192 2048; do
      PMC_DTYPE=$dt timeout 180 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_r06bf_${dt}m${m}_$i -o t -- python $R/tests/dev/pmc_tiled.py $m 4096 4096 > /tmp/pmc_r06bf_${dt}m${m}_$i.log 2>&1
    done
  done
  i=$((i+1))
done
python $R/tests/dev/pmc_collect.py r06bf /tmp $O/r06_pmc_bf16_prefill.json > /dev/null
python - <<PY
import json
d = json.load(open("$O/r06_pmc_bf16_prefill.json"))
for wl in sorted(d):
    for k, e in d[wl].items():
        if not k.startswith("tiled_kernel"): continue
        mf = (e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (e["GRBM_GUI_ACTIVE"] / 8)
        va = (4 * e["SQ_ACTIVE_INST_VALU"] / 1024) / (e["GRBM_GUI_ACTIVE"] / 8)
        print(f"{wl:14s} us_under_pmc {e['avg_us_under_pmc']:7.1f}  VALU insts {e['SQ_INSTS_VALU']/1e6:6.2f} M  MFMA insts {e['SQ_INSTS_MFMA']/1e6:5.2f} M  VALU per MFMA {e['SQ_INSTS_VALU']/e['SQ_INSTS_MFMA']:.2f}  "
              f"MFMA busy {mf:.3f}  VALU busy {va:.3f}  wait_inst {e['SQ_WAIT_INST_ANY']/e['SQ_WAVE_CYCLES']:.3f}  wait_any {e['SQ_WAIT_ANY']/e['SQ_WAVE_CYCLES']:.3f}")
PY
