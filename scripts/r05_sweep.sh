#!/bin/bash
# usage: bash scripts/r05_sweep.sh <out tag> "<ENV=val ...>" ...   -- one short bench run (8 in flight, then 1) per environment setting
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; shift; mkdir -p $out
short="--no-cpu-baseline --no-primitives --no-extra --no-check"
for E in "$@"; do
  for B in "--inflight 8" "--inflight 1" $EXTRA_MODES; do
    env $E timeout 300 python bench.py $short $B > $out/t.log 2>&1
    python - "$E $B" $out/t.log <<'PY' | tee -a $out/sweep.txt
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1]); r=j["roofline"]; h=j["lookahead_stats"]["host_ms"]
    print("%-50s fps %7.0f us/search %6.3f api_ms %7.0f prefetch_mbtree_ms %7.0f" % (sys.argv[1], j["value"], r["us_per_search"], h["api_total"], h["prefetch_mbtree"]))
except Exception as e: print(sys.argv[1],"FAILED",e)
PY
  done
done
