#!/bin/bash
# round 3, GPU session 3: full suite, speculation by position (on / off), 2-rank shard check
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s3
mkdir -p gpurun_out/$tag
( time timeout 900 python -m pytest tests -q -m gpu --durations=5 -s ) > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/$tag/summary.txt
grep -h "window shard on one GPU\|passed\|failed\|Error\|FAILED" gpurun_out/$tag/pytest.log | tail -12 | cut -c1-900 | tee -a gpurun_out/$tag/summary.txt
B="python bench.py --no-cpu-baseline --no-primitives --no-extra"
for E in "A=1" "X264HIP_NO_POSITION_CLASSES=1" "X264HIP_LA_CHUNK=64"; do
  for A in "" "--inflight 1" "--paced"; do
    env $E timeout 300 $B $A > gpurun_out/$tag/ab.log 2>&1
    python - "$E $A" gpurun_out/$tag/ab.log <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    d = j["lookahead_stats"]["device"]
    print("%-44s fps %8.1f other %8.1f us/search %6.3f solo %s | searches %d claimed %d on-demand %d | cells spec %d hits %d on-demand %d | unclaimed %.3f unused %.3f" % (
        sys.argv[1], j["value"], j.get("paced_fps") or j.get("batched_fps") or 0, j["roofline"]["us_per_search"], (j["roofline"].get("solo") or {}).get("us_per_search"),
        d["searches"], d["fields_claimed"], d["searches_on_demand"], d["cells_speculated"], d["cell_hits"], d["cells_on_demand"], d["unclaimed_field_share"], d["unused_cell_share"]))
except Exception as e:
    print("%-44s FAILED %s" % (sys.argv[1], e)); print(open(sys.argv[2]).read()[-800:])
PY
  done
done
( time X264HIP_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --shard window --no-cpu-baseline --no-primitives --no-extra --no-check --steps 1 --warmup 0 --frames 32 --inflight 1 ) > gpurun_out/$tag/shard2.log 2>&1
echo "shard2 rc=$?" | tee -a gpurun_out/$tag/summary.txt
python - gpurun_out/$tag/shard2.log <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("shard window 2 ranks (one GPU, gloo):", json.dumps(j.get("window_shard"))[:2500])
except Exception as e:
    print("shard2 FAILED", e); print(open(sys.argv[1]).read()[-2500:])
PY
