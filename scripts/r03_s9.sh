#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s9
mkdir -p gpurun_out/$tag
( time timeout 900 python -m pytest tests -q -m gpu --durations=3 ) > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/$tag/summary.txt
grep -h "passed\|failed\|Error\|FAILED" gpurun_out/$tag/pytest.log | tail -8 | cut -c1-600 | tee -a gpurun_out/$tag/summary.txt
echo "coop  $(timeout 300 python scripts/prim_bench.py me_full 2>&1 | tail -1)" | tee -a gpurun_out/$tag/summary.txt
B="python bench.py --no-cpu-baseline --no-primitives --no-extra --warmup 2 --steps 4"
for A in "" "--inflight 1" "--inflight 2" "--inflight 4" "--paced" "--paced --inflight 1"; do
    timeout 400 $B $A > gpurun_out/$tag/ab.log 2>&1
    python - "$A" gpurun_out/$tag/ab.log <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    d = j["lookahead_stats"]["device"]
    print("%-26s fps %8.1f other %8.1f | searches %d claimed %d | cells spec %d hits %d on-demand %d 2nd variants %d used %d" % (
        sys.argv[1], j["value"], j.get("paced_fps") or j.get("batched_fps") or 0,
        d["searches"], d["fields_claimed"], d["cells_speculated"], d["cell_hits"], d["cells_on_demand"], d["second_variants_speculated"], d["second_variants_used"]))
except Exception as e:
    print("%-26s FAILED %s" % (sys.argv[1], e)); print(open(sys.argv[2]).read()[-800:])
PY
done
