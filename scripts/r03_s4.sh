#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s4
mkdir -p gpurun_out/$tag
B="python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check"
for E in "X264HIP_LA_CHUNK=8" "X264HIP_LA_CHUNK=16" "X264HIP_LA_CHUNK=32" "X264HIP_LA_CHUNK=256"; do
  for A in "" "--frames 320" "--inflight 1"; do
    env $E timeout 300 $B $A > gpurun_out/$tag/ab.log 2>&1
    python - "$E $A" gpurun_out/$tag/ab.log <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    d = j["lookahead_stats"]["device"]
    print("%-44s fps %8.1f us/search %6.3f launches %d | searches %d claimed %d on-demand %d | cells spec %d hits %d on-demand %d" % (
        sys.argv[1], j["value"], j["roofline"]["us_per_search"], j["roofline"]["launches"],
        d["searches"], d["fields_claimed"], d["searches_on_demand"], d["cells_speculated"], d["cell_hits"], d["cells_on_demand"]))
except Exception as e:
    print("%-44s FAILED %s" % (sys.argv[1], e)); print(open(sys.argv[2]).read()[-800:])
PY
  done
done
for r in 2 4 8; do
  echo "cmp rows $r $(X264HIP_CMP_ROWS=$r timeout 120 python scripts/prim_bench.py sad satd 2>&1 | tail -1)" | tee -a gpurun_out/$tag/summary.txt
done
