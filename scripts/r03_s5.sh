#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s5
mkdir -p gpurun_out/$tag
B="python bench.py --no-cpu-baseline --no-primitives --no-extra --warmup 2 --steps 4"
for E in "X264HIP_LA_CHUNK=32" "X264HIP_LA_CHUNK=32 X264HIP_NO_POSITION_CLASSES=1" "X264HIP_LA_CHUNK=64" "X264HIP_LA_CHUNK=256" "X264HIP_LA_CHUNK=16"; do
  for A in "" "--frames 320" "--inflight 1" "--paced"; do
    env $E timeout 400 $B $A > gpurun_out/$tag/ab.log 2>&1
    python - "$E $A" gpurun_out/$tag/ab.log <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    d = j["lookahead_stats"]["device"]
    print("%-66s fps %8.1f other %8.1f | searches %d claimed %d on-demand %d | cells spec %d hits %d on-demand %d | unclaimed %.3f unused %.3f" % (
        sys.argv[1], j["value"], j.get("paced_fps") or j.get("batched_fps") or 0,
        d["searches"], d["fields_claimed"], d["searches_on_demand"], d["cells_speculated"], d["cell_hits"], d["cells_on_demand"], d["unclaimed_field_share"], d["unused_cell_share"]))
except Exception as e:
    print("%-66s FAILED %s" % (sys.argv[1], e)); print(open(sys.argv[2]).read()[-800:])
PY
  done
done
