#!/bin/bash
# round 3, GPU session 1: parity of the set-based search kernel, default bench, A/B against the round-2 library, primitive variants
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r03s1
mkdir -p gpurun_out/$tag
( time timeout 900 python -m pytest tests -q -m gpu -x --durations=5 ) > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/$tag/summary.txt
tail -5 gpurun_out/$tag/pytest.log | tee -a gpurun_out/$tag/summary.txt
B="python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check"
for lib in x264_amd/libx264hip_r02.so x264_amd/libx264hip.so; do
  for A in "" "--inflight 1" "--inflight 2"; do
    X264HIP_LIB=$lib timeout 300 $B $A > gpurun_out/$tag/ab.log 2>&1
    python - "$lib $A" gpurun_out/$tag/ab.log <<'PY' | tee -a gpurun_out/$tag/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print("%-50s fps %9.1f  us/search %7.3f  launch ms %7.3f" % (sys.argv[1], j["value"], j["roofline"]["us_per_search"], j["roofline"]["avg_launch_ms"]))
except Exception as e:
    print("%-50s FAILED %s" % (sys.argv[1], e))
PY
  done
done
for v in "" "1n,32" "4t,8" "4n,8" "8t,4" "8t,8" "2t,16" "4t,16" "4t,32"; do
  echo "copy [$v] $(X264HIP_COPY=$v timeout 120 python scripts/prim_bench.py device_copy 2>&1 | tail -1)" | tee -a gpurun_out/$tag/summary.txt
done
for r in 1 2 4; do
  echo "cmp rows $r $(X264HIP_CMP_ROWS=$r timeout 120 python scripts/prim_bench.py sad satd 2>&1 | tail -1)" | tee -a gpurun_out/$tag/summary.txt
done
echo "hpel $(timeout 120 python scripts/prim_bench.py hpel 2>&1 | tail -1)" | tee -a gpurun_out/$tag/summary.txt
( time timeout 900 python bench.py ) > gpurun_out/$tag/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/$tag/summary.txt
grep -h '^{' gpurun_out/$tag/bench.log | tail -1 > gpurun_out/$tag/bench.json
cut -c1-600 gpurun_out/$tag/bench.json
