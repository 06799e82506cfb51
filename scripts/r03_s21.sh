#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s21
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
    print("%-52s fps %8.1f other %8.1f | host ms frame_cost %.0f prefetch_mbtree %.0f api %.0f" % (
        sys.argv[1], j["value"], j.get("paced_fps") or j.get("batched_fps") or 0, h["frame_cost"], h["prefetch_mbtree"], h["api_total"]))
except Exception as e:
    print("%-52s FAILED %s" % (sys.argv[1], e)); print(open(sys.argv[2]).read()[-800:])
PY
}
run "A=1" ""
run "X264HIP_UPLOAD_KERNEL=1" ""
run "A=1" "--inflight 1"
run "X264HIP_UPLOAD_KERNEL=1" "--inflight 1"
run "A=1" "--inflight 2"
run "A=1" "--inflight 4"
run "A=1" "--paced"
run "A=1" "--paced --inflight 1"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/${tag}_api
timeout 600 rocprofv3 --hip-runtime-trace --stats --output-format csv -d gpurun_out/${tag}_api -o run -- python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check > gpurun_out/${tag}_api/bench.log 2>&1; echo "api rc=$?" | tee -a gpurun_out/$tag/summary.txt
