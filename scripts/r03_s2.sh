#!/bin/bash
# round 3, GPU session 2: owner-side cells (2-process shard test, 1-GPU and 2-rank bench), occupancy variants, saturated launches
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s2
mkdir -p gpurun_out/$tag
( time timeout 900 python -m pytest tests -q -m gpu -x --durations=5 -s ) > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/$tag/summary.txt
grep -h "window shard on one GPU\|passed\|failed\|Error" gpurun_out/$tag/pytest.log | tail -8 | cut -c1-900 | tee -a gpurun_out/$tag/summary.txt
B="python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check"
run() { # label, env..., -- args
  label=$1; shift
  env "$@" > /dev/null 2>&1
}
for lib in x264_amd/libx264hip_r02.so x264_amd/libx264hip.so x264_amd/libx264hip_w5.so x264_amd/libx264hip_w6.so; do
  for A in "" "--inflight 1"; do
    for CH in 64 400; do
      X264HIP_LA_CHUNK=$CH X264HIP_LIB=$lib timeout 300 $B $A > gpurun_out/$tag/ab.log 2>&1
      python - "$lib chunk=$CH $A" gpurun_out/$tag/ab.log <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print("%-60s fps %9.1f  us/search %7.3f  launch ms %7.3f searches/launch %6.0f" % (sys.argv[1], j["value"], j["roofline"]["us_per_search"], j["roofline"]["avg_launch_ms"], j["roofline"]["searches"] / max(j["roofline"]["launches"], 1)))
except Exception as e:
    print("%-60s FAILED %s" % (sys.argv[1], e))
PY
    done
  done
done
# BASELINE configs[3] on one GPU (one stream, one 250-frame GOP)
( time timeout 600 python bench.py --shard window --no-cpu-baseline --no-primitives --no-extra --no-check --steps 1 --warmup 0 --frames 32 --inflight 1 ) > gpurun_out/$tag/shard1.log 2>&1
python - gpurun_out/$tag/shard1.log <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("shard window 1 rank:", json.dumps(j.get("window_shard"))[:1500])
except Exception as e:
    print("shard1 FAILED", e); print(open(sys.argv[1]).read()[-1500:])
PY
# the same stream over two ranks sharing this GPU (gloo staging: protocol and result check, not a speed figure)
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
