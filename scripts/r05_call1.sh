#!/bin/bash
# round 5, GPU call 1: load-shape rates of the L1 / LDS (experiments/mem_rates) + the baseline search numbers of the round
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05a; mkdir -p $out
timeout 120 ./experiments/mem_rates/mem_rates > $out/mem_rates.json 2> $out/mem_rates.err; echo "mem_rates rc=$?"
cat $out/mem_rates.json
short="--no-cpu-baseline --no-primitives --no-extra --no-check"
timeout 300 python bench.py $short --inflight 8 > $out/b8.log 2>&1; echo "b8 rc=$?"
timeout 300 python bench.py $short --inflight 1 > $out/b1.log 2>&1; echo "b1 rc=$?"
python - <<'PY'
import json
for f in ("b8","b1"):
    try:
        j=json.loads([l for l in open("gpurun_out/r05a/%s.log"%f) if l.startswith("{")][-1])
        r=j["roofline"]; print(f, j["value"], r["us_per_search"], r.get("solo",{}).get("us_per_search"), j["lookahead_stats"]["device"]["unclaimed_field_share"])
    except Exception as e: print(f,"FAILED",e)
PY
