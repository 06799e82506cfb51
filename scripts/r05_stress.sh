#!/bin/bash
# usage: bash scripts/r05_stress.sh <tag> <n> [ENV=val ...]  -- n short single-stream bench runs (batched, then its paced check): any in-kernel timeout shows up as an error
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; n=$2; shift; shift; mkdir -p $out
for i in $(seq 1 $n); do
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-primitives --no-extra --inflight 1 --steps 4 --warmup 1 > $out/s.log 2>&1
  rc=$?
  python - $rc $out/s.log <<'PY' | tee -a $out/stress.txt
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1]); print("rc", sys.argv[1], "fps", j["value"], "paced", j.get("paced_fps"), "err" if "error" in json.dumps(j) else "")
except Exception as e: print("rc", sys.argv[1], "FAILED", e, open(sys.argv[2]).read()[-300:])
PY
done
