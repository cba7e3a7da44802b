#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out"; mkdir -p $O
( timeout 600 python tests/dev/glue_breakdown.py 2>&1 | grep "|" ) > $O/c6_glue.txt
( timeout 600 python tests/dev/glue_breakdown.py 70b 2>&1 | grep "|" ) > $O/c6_glue70.txt
( timeout 600 python -m pytest tests/test_gpu_decode_chain.py -m gpu -q --timeout 600 -p no:cacheprovider --tb=short 2>&1 | tail -5 ) > $O/c6_pytest.log 2>&1
( timeout 600 python bench.py --no-cpu-baseline ) > $O/c6_bench.json 2>/dev/null
cat $O/c6_glue.txt $O/c6_glue70.txt; tail -3 $O/c6_pytest.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/c6_bench.json'))
print('headline', round(d['value'],1))
for c in d.get('configs',[]):
    print(c.get('config'), c.get('mode',''), round(c.get('value',0),1), c.get('unit'), c.get('error',''))
PY
