#!/bin/bash
# usage (through gpurun): bash scripts/sweep_args.sh <tag> "<bench args>" "<bench args>" ...   one short bench run per argument set
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out/$tag
i=0
for A in "$@"; do
  i=$((i+1))
  timeout 300 python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check $A > gpurun_out/$tag/args_$i.log 2>&1
  python - "$A" gpurun_out/$tag/args_$i.log <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print("%-40s fps %9.1f  us/search %7.3f  launch ms %7.3f" % (sys.argv[1], j["value"], j["roofline"]["us_per_search"], j["roofline"]["avg_launch_ms"]))
except Exception as e:
    print("%-40s FAILED %s" % (sys.argv[1], e))
PY
done
