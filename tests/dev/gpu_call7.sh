#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out"; mkdir -p $O
: > $O/c7_waves.txt
for w in 0 7 12; do
  echo "--- force_waves=$w" >> $O/c7_waves.txt
  ( ONLY="plain,rms(stats),residual+stats_out" GPTQHIP_FORCE_VARIANT=$w timeout 300 python tests/dev/glue_breakdown.py 2>&1 | grep "|" | cut -c1-120 ) >> $O/c7_waves.txt
  ( ONLY="plain,rms(stats),residual+stats_out" GPTQHIP_FORCE_VARIANT=$w timeout 300 python tests/dev/glue_breakdown.py 70b 2>&1 | grep "|" | cut -c1-120 ) >> $O/c7_waves.txt
done
cat $O/c7_waves.txt
( timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --tb=short -x 2>&1 | tail -6 ) > $O/c7_pytest.log 2>&1
tail -4 $O/c7_pytest.log
( timeout 600 python bench.py --no-cpu-baseline ) > $O/c7_bench.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/c7_bench.json'))
print('headline', round(d['value'],1))
for c in d.get('configs',[]):
    print(c.get('config'), c.get('mode',''), round(c.get('value',0),1), c.get('unit'), c.get('error',''))
PY
