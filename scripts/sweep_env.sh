#!/bin/bash
# usage (through gpurun): bash scripts/sweep_env.sh <tag> "<VAR=val VAR2=val>" "<...>" ...   one short bench run per environment setting
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out/$tag
i=0
for E in "$@"; do
  i=$((i+1))
  env $E timeout 300 python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check ${BENCH_ARGS} > gpurun_out/$tag/run_$i.log 2>&1
  python - "$E" gpurun_out/$tag/run_$i.log <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print("%-50s fps %9.1f  us/search %7.3f  launch ms %7.3f" % (sys.argv[1], j["value"], j["roofline"]["us_per_search"], j["roofline"]["avg_launch_ms"]))
except Exception as e:
    print("%-50s FAILED %s" % (sys.argv[1], e))
PY
done
