#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s31
mkdir -p gpurun_out/$tag
( timeout 900 python -m pytest tests -q -m gpu -x ) > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/$tag/summary.txt
tail -2 gpurun_out/$tag/pytest.log | cut -c1-200 | tee -a gpurun_out/$tag/summary.txt
timeout 600 python bench.py --no-cpu-baseline --no-primitives > gpurun_out/$tag/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/$tag/summary.txt
grep -h '^{' gpurun_out/$tag/bench.log | tail -1 > gpurun_out/$tag/bench.json
python - <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json
d=json.load(open('gpurun_out/r03s31/bench.json'))
print({k:d[k] for k in ('value','steps','paced_fps')})
for k in ('configs2_4k','configs4_8k_1gpu','configs3_4k_1gpu'):
    print(k, d[k].get('value'), d[k].get('paced_fps'), d[k].get('error'))
PY
