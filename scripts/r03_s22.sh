#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s22
mkdir -p gpurun_out/$tag
: > gpurun_out/$tag/summary.txt
B="python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --warmup 2 --steps 4"
run() {
    timeout 400 env $1 $B $2 > gpurun_out/$tag/ab.log 2>&1
    python - "$1 $2" gpurun_out/$tag/ab.log <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    h = j["lookahead_stats"]["host_ms"]
    print("%-52s fps %8.1f | host ms frame_cost %.0f prefetch_mbtree %.0f api %.0f" % (
        sys.argv[1], j["value"], h["frame_cost"], h["prefetch_mbtree"], h["api_total"]))
except Exception as e:
    print("%-52s FAILED %s" % (sys.argv[1], e)); print(open(sys.argv[2]).read()[-800:])
PY
}
run "A=1" ""
run "X264HIP_UPLOAD=prio" ""
run "A=1" "--inflight 4"
run "X264HIP_UPLOAD=prio" "--inflight 4"
run "A=1" "--paced"
run "X264HIP_UPLOAD=prio" "--paced"
run "X264HIP_UPLOAD=prio" "--inflight 1"
