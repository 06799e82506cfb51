#!/bin/bash
# MB-tree lists queued and run side by side: GPU suite + A/B (X264HIP_MBT_GROUPS = 0 launch per call, 1 queued one list per launch, 8 default)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s10
mkdir -p gpurun_out/$tag
( time timeout 900 python -m pytest tests -q -m gpu --durations=3 -x ) > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/$tag/summary.txt
grep -h "passed\|failed\|Error\|FAILED" gpurun_out/$tag/pytest.log | tail -8 | cut -c1-600 | tee -a gpurun_out/$tag/summary.txt
B="python bench.py --no-cpu-baseline --no-primitives --no-extra --warmup 2 --steps 4"
run() {
    timeout 400 env $1 $B $2 > gpurun_out/$tag/ab.log 2>&1
    python - "$1 $2" gpurun_out/$tag/ab.log <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    h = j["lookahead_stats"]["host_ms"]
    print("%-44s fps %8.1f other %8.1f | host ms frame_cost %.0f prefetch_mbtree %.0f api %.0f" % (
        sys.argv[1], j["value"], j.get("paced_fps") or j.get("batched_fps") or 0, h["frame_cost"], h["prefetch_mbtree"], h["api_total"]))
except Exception as e:
    print("%-44s FAILED %s" % (sys.argv[1], e)); print(open(sys.argv[2]).read()[-800:])
PY
}
run "X264HIP_MBT_GROUPS=8" ""
run "X264HIP_MBT_GROUPS=0" ""
run "X264HIP_MBT_GROUPS=1" ""
run "X264HIP_MBT_GROUPS=4" ""
run "X264HIP_MBT_GROUPS=8 X264HIP_MBT_WGS=4" ""
run "X264HIP_MBT_GROUPS=8 X264HIP_MBT_WGS=16" ""
run "X264HIP_MBT_GROUPS=8" "--inflight 1"
run "X264HIP_MBT_GROUPS=0" "--inflight 1"
run "X264HIP_MBT_GROUPS=8" "--inflight 2"
run "X264HIP_MBT_GROUPS=8" "--paced"
run "X264HIP_MBT_GROUPS=0" "--paced"
run "X264HIP_MBT_GROUPS=8" "--paced --inflight 1"
