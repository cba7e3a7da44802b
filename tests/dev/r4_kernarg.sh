#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-configs --no-e2e"
one() { "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f tok/s  %.4f ms  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
{
echo "default:";                         one $B; one $B
echo "HIP_FORCE_DEV_KERNARG=1:";         HIP_FORCE_DEV_KERNARG=1 one $B; HIP_FORCE_DEV_KERNARG=1 one $B
echo "kernarg preload build:";           GPTQHIP_LIB=$R/tests/dev/ablate/libgptqhip_preload.so one $B
echo "preload + dev kernarg:";           HIP_FORCE_DEV_KERNARG=1 GPTQHIP_LIB=$R/tests/dev/ablate/libgptqhip_preload.so one $B
echo "no graph, default:";               one $B --no-graph
echo "no graph, HIP_FORCE_DEV_KERNARG=1:"; HIP_FORCE_DEV_KERNARG=1 one $B --no-graph
echo "bf16 default / dev kernarg:";      one $B --dtype bf16; HIP_FORCE_DEV_KERNARG=1 one $B --dtype bf16
} > gpurun_out/r4_kernarg.txt 2>&1
cat gpurun_out/r4_kernarg.txt
