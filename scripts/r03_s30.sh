#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s30
mkdir -p gpurun_out/$tag
: > gpurun_out/$tag/summary.txt
B="python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --warmup 2 --steps 6"
run() {
    timeout 400 env $1 $B $2 > gpurun_out/$tag/ab.log 2>&1
    python - "$1 $2" gpurun_out/$tag/ab.log <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print("%-60s fps %8.1f" % (sys.argv[1], j["value"]))
except Exception as e:
    print("%-60s FAILED %s" % (sys.argv[1], e)); print(open(sys.argv[2]).read()[-800:])
PY
}
run "A=1" ""
run "X264HIP_MBT_THREADS=256 X264HIP_MBT_WGS=4" ""
run "X264HIP_MBT_THREADS=256 X264HIP_MBT_WGS=8" ""
run "X264HIP_MBT_THREADS=512 X264HIP_MBT_WGS=2" ""
run "X264HIP_MBT_THREADS=512 X264HIP_MBT_WGS=4" ""
run "X264HIP_MBT_THREADS=256 X264HIP_MBT_WGS=2" ""
run "A=1" ""
run "X264HIP_MBT_THREADS=256 X264HIP_MBT_WGS=4" "--inflight 1"
run "A=1" "--inflight 1"
