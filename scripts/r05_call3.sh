#!/bin/bash
# round 5, GPU call: whole GPU suite with the new search build + the full-size goldens, request classes of the bench, MB-tree's share
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05c; mkdir -p $out
( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=6 ) > $out/pytest.log 2>&1; echo "suite rc=$?"; tail -12 $out/pytest.log
short="--no-cpu-baseline --no-primitives --no-extra --no-check"
X264HIP_TRACE_CLASSES=1 timeout 300 python bench.py $short --inflight 1 > $out/classes.log 2>&1; grep -h "L0 d1\|classes\|(1,0)" $out/classes.log | tail -6
for E in "A=0" "X264HIP_BENCH_CFG=mb_tree=0" "X264HIP_MBT_WGS=1" "X264HIP_MBT_WGS=4"; do
  for B in "--inflight 8" "--inflight 1"; do
    env $E timeout 300 python bench.py $short $B > $out/t.log 2>&1
    python - "$E $B" <<'PY'
import json,sys
try:
    j=json.loads([l for l in open("gpurun_out/r05c/t.log") if l.startswith("{")][-1]); r=j["roofline"]; h=j["lookahead_stats"]["host_ms"]
    print(sys.argv[1], "fps %.0f us/search %.3f host: %s" % (j["value"], r["us_per_search"], h))
except Exception as e: print(sys.argv[1],"FAILED",e)
PY
  done
done
