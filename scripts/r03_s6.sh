#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s6
mkdir -p gpurun_out/$tag
( time timeout 900 python -m pytest tests -q -m gpu --durations=5 ) > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/$tag/summary.txt
grep -h "passed\|failed\|Error\|FAILED" gpurun_out/$tag/pytest.log | tail -12 | cut -c1-600 | tee -a gpurun_out/$tag/summary.txt
echo "coop  $(timeout 300 python scripts/prim_bench.py me_full 2>&1 | tail -1)" | tee -a gpurun_out/$tag/summary.txt
echo "scalar $(X264HIP_ME_FULL_SCALAR=1 timeout 300 python scripts/prim_bench.py me_full 2>&1 | tail -1)" | tee -a gpurun_out/$tag/summary.txt
( time timeout 900 python bench.py ) > gpurun_out/$tag/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/$tag/summary.txt
grep -h '^{' gpurun_out/$tag/bench.log | tail -1 > gpurun_out/$tag/bench.json
python - <<'PY' | tee -a gpurun_out/r03s6/summary.txt
import json
j = json.load(open("gpurun_out/r03s6/bench.json"))
print("value", j["value"], "paced", j.get("paced_fps"), "solo", (j["roofline"].get("solo") or {}).get("us_per_search"))
for k in ("configs2_4k", "configs3_4k_1gpu", "configs4_8k_1gpu"):
    print(k, json.dumps(j.get(k))[:700])
print("primitives", json.dumps(j.get("primitives")))
print("cpu", json.dumps(j.get("cpu_baseline"))[:300])
PY
tail -3 gpurun_out/$tag/bench.log | cut -c1-300
