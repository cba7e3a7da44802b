#!/bin/bash
# Run ON THE GPU BOX (via gpurun): SQ / TCC counter passes for the decode ops and the prefill kernel.
#   usage: pmc_passes.sh <tag>        -> gpurun_out/<tag>_pmc.json (+ <tag>_pmc_counters.txt = rocprofv3 -L)
# One counter group per rocprofv3 run, --kernel-trace only beside --pmc (MI355X_MICROARCH.md "rocprofv3 PMC slots": 8 SQ slots, FETCH_SIZE
# and WRITE_SIZE cannot share a pass).  Round 5's collector never ran an SQ pass at all, which is why profiles/r03..r05_pmc_summary.json
# carry sq_per_launch = null.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/${TAG}_pmc_counters.txt 2>&1
GROUPS_SQ=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
  "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
  "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"
  "GRBM_GUI_ACTIVE GRBM_COUNT"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
i=0
for ctr in "${GROUPS_SQ[@]}"; do
  for wl in decode tiled8192 tiled128; do
    case $wl in
      decode)    cmd="python $R/tests/dev/pmc_decode.py" ;;
      tiled8192) cmd="python $R/tests/dev/pmc_tiled.py 8192 4096 4096" ;;
      tiled128)  cmd="python $R/tests/dev/pmc_tiled.py 128 4096 11008" ;;
    esac
    timeout 180 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${TAG}_${wl}_$i -o t -- $cmd > /tmp/pmc_${TAG}_${wl}_$i.log 2>&1
  done
  i=$((i+1))
done
python $R/tests/dev/pmc_collect.py $TAG /tmp $O/${TAG}_pmc.json
echo pmc_done
