#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s7
mkdir -p gpurun_out/$tag
timeout 600 python -m pytest tests/test_gpu_primitives.py -q -m gpu -k "me_search or me_full" 2>&1 | tail -2 | tee gpurun_out/$tag/summary.txt
echo "coop  $(timeout 300 python scripts/prim_bench.py me_full 2>&1 | tail -1)" | tee -a gpurun_out/$tag/summary.txt
B="python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --warmup 2 --steps 4"
for E in "A=1" "X264HIP_CELL_RATE=30" "X264HIP_CELL_RATE=50" "X264HIP_CELL_RATE=80" "X264HIP_FIELD_RATE=45" "X264HIP_FIELD_RATE=45 X264HIP_CELL_RATE=50"; do
  for A in "" "--inflight 12"; do
    env $E timeout 400 $B $A > gpurun_out/$tag/ab.log 2>&1
    python - "$E $A" gpurun_out/$tag/ab.log <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    d = j["lookahead_stats"]["device"]
    print("%-66s fps %8.1f | searches %d claimed %d on-demand %d | cells spec %d hits %d on-demand %d | unclaimed %.3f unused %.3f" % (
        sys.argv[1], j["value"],
        d["searches"], d["fields_claimed"], d["searches_on_demand"], d["cells_speculated"], d["cell_hits"], d["cells_on_demand"], d["unclaimed_field_share"], d["unused_cell_share"]))
except Exception as e:
    print("%-66s FAILED %s" % (sys.argv[1], e)); print(open(sys.argv[2]).read()[-800:])
PY
  done
done
